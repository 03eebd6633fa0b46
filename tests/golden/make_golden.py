"""Regenerate tests/golden/*.json from the ORACLE (oracle/).  These are regression fixtures of the CPU
restatement — the reference itself (Rust + un-vendored Plonky3) cannot run in this environment, so no
fixture here comes from the reference; the only reference-pinned values are the VM counts asserted in
tests/test_oracle_cpu.py::test_reference_pinned_vm_counts.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import valida_amd as va  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def make(n, name):
    """n: Fibonacci loop bound, or the name of one of the reference's other pinned programs / ("alu", iters)."""
    rc = va.poseidon_round_constants()
    w = va.Workload.fib(n) if isinstance(n, int) else va.Workload.alu(n[1]) if isinstance(n, tuple) else va.Workload.named(n)
    mt, prep = w.main_traces(), w.preprocessed()
    res = po.prove_basic(mt, prep[0][1], prep[1][1], rc, debug_check=True)
    assert po.verify_basic(prep[0][1], prep[1][1], res.words, rc) is None
    out = {
        "n": n,
        "poseidon_seed": "0x56414C494441",
        "cycles": w.cycles, "mem_ops": w.mem_ops, "add_ops": w.add_ops, "result": w.result,
        "traces_sha256": hashlib.sha256(b"".join(m.tobytes() for m in mt)).hexdigest(),
        "commitments": [int(x) for x in res.words[2:26]],
        "transcript": [int(x) for x in res.transcript],
        "proof_words": int(res.words.size),
        "proof_sha256": hashlib.sha256(res.bytes()).hexdigest(),
    }
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(out, f, indent=1)
    print(name, out["proof_sha256"])


if __name__ == "__main__":
    make(25, "fib25_oracle.json")
    make(582, "fib582_oracle.json")
    for prog in ("left_imm_ops", "signed_inequality", "loadfp", "static_data"):  # basic/tests/test_prover.rs:190-402, test_static_data.rs:31-59
        make(prog, prog + "_oracle.json")
    make(("alu", 100), "alu100_oracle.json")
