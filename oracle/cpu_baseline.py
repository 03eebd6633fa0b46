"""ORACLE — TEST INFRASTRUCTURE ONLY.  bench.py's cpu_baseline leg, run as a process of its own:

    python -m oracle.cpu_baseline <fib loop bound> [scalar]

proves ONE segment with the oracle (fast mode unless `scalar`) on every host core and prints one JSON line {"seconds", "cores", "fast",
"sha256", "words"}.  A separate process because the measurement must not share its heap, its OpenMP pool or its cores with the HIP runtime
and the prover threads of the benchmark process (measured on the MI355X box: 13.3 s inside bench.py's process against 4.1 s alone)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import valida_amd as va  # the workload generator (host code of the library: VM + generate_trace restatement); no device is touched
    from oracle import pyoracle as po

    n = int(sys.argv[1])
    fast = not (len(sys.argv) > 2 and sys.argv[2] == "scalar")
    mmcs = int(os.environ.get("ORACLE_MMCS", "0"))
    w = va.Workload.fib(n)
    prep = w.preprocessed()
    rc = va.poseidon_round_constants()
    if mmcs:
        po.set_mmcs_hash(1, rc)
    po.set_fast(fast)
    if fast and os.environ.get("ORACLE_KEEP_HEAP", "1") != "0":
        po.keep_heap()
    res = po.prove_basic(w.main_traces(), prep[0][1], prep[1][1], rc)
    print(json.dumps({"seconds": res.seconds, "cores": po.usable_cores(), "fast": fast, "sha256": hashlib.sha256(res.bytes()).hexdigest(), "words": int(res.words.size)}), flush=True)


if __name__ == "__main__":
    main()
