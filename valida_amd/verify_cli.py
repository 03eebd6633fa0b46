"""`valida verify` for proofs of this backend (the reference's CLI: basic/src/bin/valida.rs, verify sub-command): Machine::verify of the
product — vgpu_verify — over a proof file.  Host-only: runs anywhere the library loads, no device, no oracle.

    python -m valida_amd.verify_cli PROOF --program fib --n 25 [--log-blowup 1] [--queries 40] [--pow-bits 8] [--poseidon-mmcs]
                                          [--constants splitmix|cli] [--try-conventions]

PROOF: the CBOR image the reference writes (`ciborium::into_writer(&proof, ..)`, either setting of the two encoding switches) or raw
little-endian VPF1 words.  The program selects the preprocessed traces (ROM, range table) whose commitment the verifier recomputes
(basic/src/lib.rs:791-804): fib N | alu ITERS | left_imm_ops | signed_inequality | loadfp | static_data.  Exit status 0 = accepted.

--constants cli: the Poseidon round constants of the reference's CLI (Pcg64 from Seeder::from("validia seed"), basic/src/bin/valida.rs:364-365;
valida_amd/cli_constants.py) instead of this repository's SplitMix64 set.
--try-conventions: FIRST CONTACT with a file written by the real `valida prove`.  Everything the absent crates decide is recall (SURVEY.md
Appendix B); this mode decodes the image, reports which encodings it met, then runs the verifier under every combination of the open
switches — bare integers canonical / Montgomery, round constants raw-Montgomery / canonical, SipRng start 0x13 / 0xff / 0xee, final
polynomial observed or not — and prints the first accepted combination, or each combination's rejection (the stage that failed tells which
convention is off: "proof of work" / a FRI root = transcript or constants; a Merkle path = digest mapping; constraints = opened values)."""
import argparse
import itertools
import sys

import numpy as np

FORMS = {1: 'field elements as {"value": <Montgomery word>}', 2: "field elements as bare integers", 4: "digests as Hash { value, _marker }", 8: "digests as plain [Val; 8]"}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("proof")
    ap.add_argument("--program", default="fib")
    ap.add_argument("--n", type=int, default=25)
    ap.add_argument("--log-blowup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=40)
    ap.add_argument("--pow-bits", type=int, default=8)
    ap.add_argument("--poseidon-mmcs", action="store_true", help="the proof was committed with the Poseidon-16 MMCS (hash_kind 1)")
    ap.add_argument("--constants", choices=["splitmix", "cli"], default="splitmix")
    ap.add_argument("--try-conventions", action="store_true")
    args = ap.parse_args(argv)

    import valida_amd as va

    raw = open(args.proof, "rb").read()
    is_words = raw[:4] == (0x31465056).to_bytes(4, "little") and len(raw) % 4 == 0
    w = va.Workload.fib(args.n) if args.program == "fib" else va.Workload.alu(args.n) if args.program == "alu" else va.Workload.named(args.program)
    hash_kind = va.HASH_POSEIDON16 if args.poseidon_mmcs else va.HASH_KECCAK256
    machine = va.Machine.basic()

    def check(words, rc, observe_final_poly=False):
        cfg = dict(log_blowup=args.log_blowup, num_queries=args.queries, pow_bits=args.pow_bits, hash_kind=hash_kind, observe_final_poly=observe_final_poly)
        prep_commit = va.host_commit_root([m for _, m in w.preprocessed()], rc, log_blowup=args.log_blowup, hash_kind=hash_kind)
        return va.verify(machine, rc, words, prep_commit, **cfg)

    if not args.try_conventions:
        try:
            words = np.frombuffer(raw, dtype="<u4").astype(np.uint32) if is_words else va.proof_from_cbor(raw)
        except va.VgpuError as e:
            print("REJECTED: " + str(e))
            return 1
        msg = check(words, va.poseidon_round_constants(source=args.constants))
        print("accepted" if msg is None else "REJECTED: " + msg)
        return 0 if msg is None else 1

    # ---- first contact ----
    decoded = {}
    if is_words:
        decoded[False] = np.frombuffer(raw, dtype="<u4").astype(np.uint32)
        print("input: raw VPF1 words")
    else:
        for bare_monty in (False, True):
            try:
                words, seen = va.proof_from_cbor_ex(raw, bare_monty)
            except va.VgpuError as e:
                print("the CBOR image does not decode: %s" % e)
                return 1
            if not bare_monty:
                print("the CBOR image decodes (%d proof words); it holds: %s" % (words.size, "; ".join(v for k, v in FORMS.items() if seen & k)))
            if bare_monty and not seen & 2:
                continue  # no bare integer in the image: nothing to reinterpret
            decoded[bare_monty] = words
    const_sets = [("splitmix (this repository's)", {"source": "splitmix"})] if args.constants == "splitmix" else []
    const_sets += [("cli, raw_monty=%s, sip_adj0=%#x" % (rm, adj), {"source": "cli", "raw_monty": rm, "sip_adj0": adj}) for rm in (True, False) for adj in (0x13, 0xFF, 0xEE)]
    failures = []
    for (bare_monty, words), (cname, ckw), ofp in itertools.product(decoded.items(), const_sets, (False, True)):
        label = "bare integers %s | constants %s | final polynomial %s" % ("Montgomery" if bare_monty else "canonical", cname, "observed" if ofp else "not observed")
        msg = check(words, va.poseidon_round_constants(**ckw), ofp)
        if msg is None:
            print("ACCEPTED under: " + label)
            return 0
        failures.append((label, msg))
    print("REJECTED under every combination of the open conventions:")
    for label, msg in failures:
        print("  %-110s -> %s" % (label, msg))
    return 1


if __name__ == "__main__":
    sys.exit(main())
