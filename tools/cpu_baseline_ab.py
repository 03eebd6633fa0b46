"""Times the oracle's fast mode on the headline segment (what bench.py's cpu_baseline leg runs), optionally with per-phase clocks:
    [ORACLE_TIMING=1] [ORACLE_KEEP_HEAP=0] python tools/cpu_baseline_ab.py [n = 149794]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import valida_amd as va  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149794
w = va.Workload.fib(n)
prep = w.preprocessed()
rc = va.poseidon_round_constants()
po.set_fast(True)
if os.environ.get("ORACLE_KEEP_HEAP", "1") != "0":
    po.keep_heap()
t = time.time()
b = po.prove_basic(w.main_traces(), prep[0][1], prep[1][1], rc)
print("fast mode: %.2f s (%.2f s inside prove) on %d cores, sha256 %s" % (time.time() - t, b.seconds, po.usable_cores(), hashlib.sha256(b.bytes()).hexdigest()), flush=True)
