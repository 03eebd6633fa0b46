"""Standalone CPU acceptance check — TEST INFRASTRUCTURE: the restated Machine::verify
(basic/src/lib.rs:677-1064, machine/src/verify.rs:11-107) over a proof file, independent of the GPU prover.

    python -m oracle.verify_tool PROOF --program fib --n 25 [--log-blowup 1] [--queries 40] [--pow-bits 8]

PROOF: little-endian VPF1 words (`Proof.bytes()`) or the CBOR image (`Proof.cbor()`, either variant).  The program selects
the preprocessed traces (ROM, range table) the verifier commits to: fib N | alu ITERS | left_imm_ops | signed_inequality |
loadfp | static_data.  Exit status 0 = accepted."""
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
    args = ap.parse_args(argv)

    import valida_amd as va  # workload VM (host code) for the preprocessed traces and the Poseidon constants
    from oracle import cbor_ref, pyoracle as po

    raw = open(args.proof, "rb").read()
    if raw[:4] == (0x31465056).to_bytes(4, "little"):
        words = np.frombuffer(raw, dtype="<u4").astype(np.uint32)
    else:
        words = np.array(cbor_ref.words_from_model(cbor_ref.decode(raw)), dtype=np.uint32)
    w = va.Workload.fib(args.n) if args.program == "fib" else va.Workload.alu(args.n) if args.program == "alu" else va.Workload.named(args.program)
    prep = w.preprocessed()
    msg = po.verify_basic(prep[0][1], prep[1][1], words, va.poseidon_round_constants(), log_blowup=args.log_blowup, num_queries=args.queries,
                          pow_bits=args.pow_bits)
    print("accepted" if msg is None else "REJECTED: " + msg)
    return 0 if msg is None else 1


if __name__ == "__main__":
    sys.exit(main())
