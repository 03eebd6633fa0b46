"""`valida verify` for proofs of this backend (the reference's CLI: basic/src/bin/valida.rs, verify sub-command): Machine::verify of the
product — vgpu_verify — over a proof file.  Host-only: runs anywhere the library loads, no device, no oracle.

    python -m valida_amd.verify_cli PROOF --program fib --n 25 [--log-blowup 1] [--queries 40] [--pow-bits 8] [--poseidon-mmcs]

PROOF: the CBOR image the reference writes (`ciborium::into_writer(&proof, ..)`, either setting of the two encoding switches) or raw
little-endian VPF1 words.  The program selects the preprocessed traces (ROM, range table) whose commitment the verifier recomputes
(basic/src/lib.rs:791-804): fib N | alu ITERS | left_imm_ops | signed_inequality | loadfp | static_data.  Exit status 0 = accepted."""
import argparse
import sys

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("proof")
    ap.add_argument("--program", default="fib")
    ap.add_argument("--n", type=int, default=25)
    ap.add_argument("--log-blowup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=40)
    ap.add_argument("--pow-bits", type=int, default=8)
    ap.add_argument("--poseidon-mmcs", action="store_true", help="the proof was committed with the Poseidon-16 MMCS (hash_kind 1)")
    args = ap.parse_args(argv)

    import valida_amd as va

    raw = open(args.proof, "rb").read()
    try:
        words = np.frombuffer(raw, dtype="<u4").astype(np.uint32) if raw[:4] == (0x31465056).to_bytes(4, "little") and len(raw) % 4 == 0 else va.proof_from_cbor(raw)
    except va.VgpuError as e:
        print("REJECTED: " + str(e))
        return 1
    w = va.Workload.fib(args.n) if args.program == "fib" else va.Workload.alu(args.n) if args.program == "alu" else va.Workload.named(args.program)
    rc = va.poseidon_round_constants()
    cfg = dict(log_blowup=args.log_blowup, num_queries=args.queries, pow_bits=args.pow_bits, hash_kind=va.HASH_POSEIDON16 if args.poseidon_mmcs else va.HASH_KECCAK256)
    prep_commit = va.host_commit_root([m for _, m in w.preprocessed()], rc, log_blowup=args.log_blowup, hash_kind=cfg["hash_kind"])
    msg = va.verify(va.Machine.basic(), rc, words, prep_commit, **cfg)
    print("accepted" if msg is None else "REJECTED: " + msg)
    return 0 if msg is None else 1


if __name__ == "__main__":
    sys.exit(main())
