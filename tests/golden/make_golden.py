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


def make(n, name, log_blowup=1, out_dir=HERE, debug_check=True, hash_kind=0):
    """n: Fibonacci loop bound, or the name of one of the reference's other pinned programs / ("alu", iters).
    hash_kind 1: the Poseidon-16 MMCS (BASELINE.json's north-star Merkle variant) instead of the reference's Keccak-256."""
    rc = va.poseidon_round_constants()
    po.set_mmcs_hash(hash_kind, rc)
    w = va.Workload.fib(n) if isinstance(n, int) else va.Workload.alu(n[1]) if isinstance(n, tuple) else va.Workload.named(n)
    mt, prep = w.main_traces(), w.preprocessed()
    res = po.prove_basic(mt, prep[0][1], prep[1][1], rc, log_blowup=log_blowup, debug_check=debug_check)
    assert po.verify_basic(prep[0][1], prep[1][1], res.words, rc, log_blowup=log_blowup) is None
    out = {
        "log_blowup": log_blowup, "hash_kind": hash_kind, "oracle_seconds": round(res.seconds, 2), "host_cores": os.cpu_count(),
        "n": n,
        "poseidon_seed": "0x56414C494441",
        "cycles": w.cycles, "mem_ops": w.mem_ops, "add_ops": w.add_ops, "result": w.result,
        "traces_sha256": hashlib.sha256(b"".join(m.tobytes() for m in mt)).hexdigest(),
        "commitments": [int(x) for x in res.words[2:26]],
        "transcript": [int(x) for x in res.transcript],
        "proof_words": int(res.words.size),
        "proof_sha256": hashlib.sha256(res.bytes()).hexdigest(),
    }
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, name), "w") as f:
        json.dump(out, f, indent=1)
    po.set_mmcs_hash(0)
    print(name, out["proof_sha256"])


def make_proof_file(out_dir=HERE):
    """fib25_q4_proof.cbor: prove_fibonacci proved by the oracle with 4 queries, as the CBOR image the product writes (canonical fields,
    plain digests) — the committed input of `python -m valida_amd.verify_cli` (tests/test_machine_verify_cpu.py)."""
    rc = va.poseidon_round_constants()
    w = va.Workload.fib(25)
    mt, prep = w.main_traces(), w.preprocessed()
    res = po.prove_basic(mt, prep[0][1], prep[1][1], rc, num_queries=4)
    blob = va.proof_cbor(res.words, va.CBOR_CANONICAL_FIELDS | va.CBOR_PLAIN_DIGESTS)
    with open(os.path.join(out_dir, "fib25_q4_proof.cbor"), "wb") as f:
        f.write(blob)
    print("fib25_q4_proof.cbor", hashlib.sha256(blob).hexdigest())


def make_full(out_dir, which):
    """BASELINE.json's full-size configurations (C2, C4, C3): minutes of oracle time on a many-core host — run once on the
    GPU box's host through gpurun (`python tests/golden/make_golden.py --full gpurun_out/golden c2 c4 c3`), then commit the
    files under tests/golden/."""
    if "c2" in which:
        make(149794, "full_c2_fib149794.json", 1, out_dir, debug_check=False)
    if "c4" in which:
        make(("alu", 116507), "full_c4_alu116507.json", 1, out_dir, debug_check=False)
    if "c3" in which:
        make(599183, "full_c3_fib599183_blowup4.json", 2, out_dir, debug_check=False)
    if "c2p" in which:  # the configuration `bench.py --mmcs poseidon` times
        make(149794, "full_c2_fib149794_poseidon.json", 1, out_dir, debug_check=False, hash_kind=1)
    if "c4p" in which:  # BASELINE.json configs[3] to the letter: the ALU / range-check heavy program with the Poseidon Merkle tree (`--workload c4 --mmcs poseidon`)
        make(("alu", 116507), "full_c4_alu116507_poseidon.json", 1, out_dir, debug_check=False, hash_kind=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--full":
        make_full(sys.argv[2], sys.argv[3:] or ["c2", "c4", "c3"])
        sys.exit(0)
    make(25, "fib25_oracle.json")
    make(582, "fib582_oracle.json")
    for prog in ("left_imm_ops", "signed_inequality", "loadfp", "static_data"):  # basic/tests/test_prover.rs:190-402, test_static_data.rs:31-59
        make(prog, prog + "_oracle.json")
    make(("alu", 100), "alu100_oracle.json")
    make_proof_file()
