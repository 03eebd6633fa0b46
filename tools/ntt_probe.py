"""Times the LDE kernels of one commit (dev tool): python tools/ntt_probe.py [log_h] [width]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import valida_amd as va

log_h = int(sys.argv[1]) if len(sys.argv) > 1 else 22
w = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rc = va.poseidon_round_constants()
p = va.Prover(va.Machine.basic(), rc)
m = np.random.default_rng(1).integers(0, va.P, size=(1 << log_h, w), dtype=np.uint32)
t = p.upload(m)
for _ in range(2):
    p.commit_batches([t])
p.set_profiling(True)
for _ in range(5):
    p.commit_batches([t])
prof = p.profile()
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print("%-18s launches %3d  ms/launch %.4f  GB/s %.0f" % (k, v[0], v[1] / v[0], v[2] / (v[1] * 1e-3) / 1e9 if v[1] else 0))
