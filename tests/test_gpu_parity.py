"""GPU parity tests: every stage of the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs (bit-exact: this is integer arithmetic mod p), then the whole proof.

Run on the MI355X box with `pytest -m gpu`.
"""
import time

import numpy as np
import pytest

import valida_amd as va
from conftest import first_mismatch
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
P = va.P


def rand_matrix(rng, h, w):
    return rng.integers(0, P, size=(h, w), dtype=np.uint32)


# ---- LDE: coset_lde_batch + bit_reverse_rows (App. B3/B4) --------------------------------------------
@pytest.mark.parametrize("log_h,w", [(0, 3), (1, 2), (3, 5), (6, 7), (10, 3), (12, 2), (13, 3), (14, 2), (16, 1)])
def test_lde_matches_oracle(prover, log_h, w):
    rng = np.random.default_rng(1000 + log_h)
    m = rand_matrix(rng, 1 << log_h, w)
    pd = prover.commit_batches([prover.upload(m)])
    got = pd.lde(0)
    want = po.committed_lde(m, 1, 31)
    assert first_mismatch(got, want) is None
    assert first_mismatch(pd.root, po.commit_root([m])) is None


def test_lde_shifted_matches_oracle(prover):
    rng = np.random.default_rng(7)
    m = rand_matrix(rng, 1 << 9, 10)
    shift = 31 * 31 % P
    pd = prover.commit_batches([prover.upload(m)], coset_shifts=[shift])
    want = po.committed_lde(m, 1, pow(31, P - 2, P))  # 31 / 31^2
    assert first_mismatch(pd.lde(0), want) is None
    assert first_mismatch(pd.root, po.commit_root([m], shifts=[shift])) is None


# ---- MMCS: mixed heights, injection, wide rows (several Keccak blocks), commit order ----------------
def test_mixed_height_commit_root(prover):
    rng = np.random.default_rng(11)
    shapes = [(1 << 8, 51), (1 << 5, 1), (1 << 9, 14), (1 << 7, 16), (1, 16), (1 << 8, 34), (1, 79), (1 << 5, 33), (1 << 9, 68)]
    mats = [rand_matrix(rng, h, w) for h, w in shapes]
    pd = prover.commit_batches([prover.upload(m) for m in mats])
    assert first_mismatch(pd.root, po.commit_root(mats)) is None


def test_commit_root_across_the_lane_pair_threshold(prover):
    """Layers of at most 32768 nodes are hashed by lane pairs (keccak_pair.hpp), larger ones one node per thread: a tree whose injections
    land on both sides, with row widths that end exactly on / one before / one after a 34-word Keccak block."""
    rng = np.random.default_rng(13)
    shapes = [(1 << 16, 2), (1 << 15, 35), (1 << 14, 69), (1 << 13, 34), (1 << 12, 33), (1 << 11, 67), (1 << 10, 68), (1 << 9, 1), (1 << 14, 1)]
    mats = [rand_matrix(rng, h, w) for h, w in shapes]
    pd = prover.commit_batches([prover.upload(m) for m in mats])
    assert first_mismatch(pd.root, po.commit_root(mats)) is None


def test_single_row_batch_root(prover):
    rng = np.random.default_rng(12)
    mats = [rand_matrix(rng, 1, w) for w in (3, 40, 7)]
    pd = prover.commit_batches([prover.upload(m) for m in mats])
    assert first_mismatch(pd.root, po.commit_root(mats)) is None


# ---- permutation traces (machine/src/chip.rs:121-208) -----------------------------------------------
@pytest.mark.parametrize("chip", [0, 2, 3, 5, 12, 13])
def test_perm_trace_matches_oracle(prover, fib25, chip):
    rng = np.random.default_rng(100 + chip)
    ch = rng.integers(0, P, size=15, dtype=np.uint32)
    main = fib25.main_trace(chip)
    got, cs = prover.generate_permutation_trace(chip, prover.upload(main), ch)
    want = po.perm_trace(chip, main, ch)
    assert first_mismatch(got, want) is None
    assert first_mismatch(cs, want[-1, -5:]) is None


def test_perm_trace_random_rows(prover):
    # random (unsatisfying) rows still define the same function; exercises every column weight
    rng = np.random.default_rng(5)
    for chip, w in [(3, 16), (7, 28), (8, 45), (10, 79)]:
        main = rand_matrix(rng, 1 << 6, w)
        ch = rng.integers(0, P, size=15, dtype=np.uint32)
        got, _ = prover.generate_permutation_trace(chip, prover.upload(main), ch)
        assert first_mismatch(got, po.perm_trace(chip, main, ch)) is None


def test_perm_trace_long_scan(prover):
    # heights beyond one scan block and beyond one sums block
    rng = np.random.default_rng(6)
    main = rand_matrix(rng, 1 << 13, 14)
    ch = rng.integers(0, P, size=15, dtype=np.uint32)
    got, _ = prover.generate_permutation_trace(2, prover.upload(main), ch)
    assert first_mismatch(got, po.perm_trace(2, main, ch)) is None


# ---- FRI fold (App. B10) ------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n", [2, 5, 12, 14])
def test_fri_fold_matches_oracle(prover, log_n):
    rng = np.random.default_rng(200 + log_n)
    f = rng.integers(0, P, size=(1 << log_n, 5), dtype=np.uint32)
    beta = rng.integers(0, P, size=5, dtype=np.uint32)
    assert first_mismatch(prover.fri_fold(f, beta), po.fri_fold(f, beta)) is None


# ---- the whole proof ----------------------------------------------------------------------------------
def _prove_both(prover, w, rc):
    mt = w.main_traces()
    prep = w.preprocessed()
    dmain = [prover.upload(m) for m in mt]
    dprep = [(c, prover.upload(m)) for c, m in prep]
    proof = prover.prove(dmain, dprep, debug=True)
    ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc)
    return proof, ref, prep


def test_fib25_proof_stages_and_bytes(prover, fib25, rc):
    proof, ref, prep = _prove_both(prover, fib25, rc)
    names = ["prep_root"] * 8 + ["perm_challenges"] * 15 + ["alpha"] * 5 + ["zeta"] * 5
    # stage by stage, in transcript order, so the first failing stage is the one reported
    assert first_mismatch(proof.transcript[:8], ref.transcript[:8]) is None, "preprocessed commitment"
    assert first_mismatch(proof.words[2:10], ref.words[2:10]) is None, "main commitment"
    assert first_mismatch(proof.transcript[8:23], ref.transcript[8:23]) is None, "perm challenges"
    for chip in range(va.NUM_CHIPS):
        assert first_mismatch(proof.debug_perm_trace(chip), ref.perm_trace(chip)) is None, "perm trace of chip %d" % chip
    assert first_mismatch(proof.words[10:18], ref.words[10:18]) is None, "perm commitment"
    assert first_mismatch(proof.transcript[23:28], ref.transcript[23:28]) is None, "alpha"
    for chip in range(va.NUM_CHIPS):
        assert first_mismatch(proof.debug_quotient(chip), ref.quotient(chip)) is None, "quotient chunks of chip %d (%s)" % (chip, va.CHIP_NAMES[chip])
    assert first_mismatch(proof.words[18:26], ref.words[18:26]) is None, "quotient commitment"
    assert first_mismatch(proof.transcript[28:33], ref.transcript[28:33]) is None, "zeta"
    assert first_mismatch(proof.words, ref.words) is None, "proof words"
    assert proof.bytes() == ref.bytes()
    # and the oracle's restated Machine::verify accepts the GPU proof
    assert po.verify_basic(prep[0][1], prep[1][1], proof.words, rc) is None


def test_fib_medium_proof_bytes(prover, rc):
    # cpu 2^12, mem 2^14: exercises the two-pass NTT, multi-block scans and 14+ FRI layers
    w = va.Workload.fib(580)
    assert w.cpu_height == 1 << 12
    proof, ref, prep = _prove_both(prover, w, rc)
    assert first_mismatch(proof.words, ref.words) is None
    assert po.verify_basic(prep[0][1], prep[1][1], proof.words, rc) is None


@pytest.mark.parametrize("name", ["fib25_oracle.json", "fib582_oracle.json", "left_imm_ops_oracle.json", "signed_inequality_oracle.json",
                                  "loadfp_oracle.json", "static_data_oracle.json", "alu100_oracle.json"])
def test_gpu_proof_matches_committed_golden_fixture(prover, name):
    import hashlib
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", name)) as f:
        g = json.load(f)
    n = g["n"]
    w = va.Workload.fib(n) if isinstance(n, int) else va.Workload.alu(n[1]) if isinstance(n, list) else va.Workload.named(n)
    mt, prep = w.main_traces(), w.preprocessed()
    assert hashlib.sha256(b"".join(m.tobytes() for m in mt)).hexdigest() == g["traces_sha256"]
    proof = prover.prove([prover.upload(m) for m in mt], [(c, prover.upload(m)) for c, m in prep])
    assert [int(x) for x in proof.words[2:26]] == g["commitments"]
    assert [int(x) for x in proof.transcript] == g["transcript"]
    assert hashlib.sha256(proof.bytes()).hexdigest() == g["proof_sha256"]


# Every A / B switch of the library that selects ANOTHER kernel for the same step ("Environment switches") must give the oracle's proof as
# well: the alternatives stay in the tree as the A / B baselines of profiles/r0*_ab_*, and several are the path of run-time captured AIRs or of the
# sharded prover.  The switches are read once per process, so each setting proves in a process of its own: fib(582) (cpu 2^12, mem 2^14: the big-matrix
# kernels and the one-tile ones both run; reduced openings above and below the 1024-row threshold) against the committed fixture of the oracle's proof.
AB_SETTINGS = ["VGPU_REDUCE_ROWS=1", "VGPU_REDUCE_ROWS=2", "VGPU_REDUCE_ROWS=4", "VGPU_PERM_NATIVE=0", "VGPU_QUOT_PER_POINT=0", "VGPU_LDE_FUSED=0", "VGPU_KECCAK_PAIRS=0",
               "VGPU_KECCAK_LEVELS=0"]


@pytest.mark.parametrize("setting", AB_SETTINGS)
def test_every_ab_switch_gives_the_oracles_proof(setting):
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import hashlib, json, sys; sys.path.insert(0, %r); import valida_amd as va\n"
            "g = json.load(open(%r)); w = va.Workload.fib(g['n']); mt, prep = w.main_traces(), w.preprocessed()\n"
            "p = va.Prover(va.Machine.basic(), va.poseidon_round_constants(), device=0)\n"
            "pr = p.prove([p.upload(m) for m in mt], [(c, p.upload(m)) for c, m in prep])\n"
            "print(json.dumps({'sha': hashlib.sha256(pr.bytes()).hexdigest(), 'commitments': [int(x) for x in pr.words[2:26]]}))\n") % (
                root, os.path.join(root, "tests", "golden", "fib582_oracle.json"))
    k, v = setting.split("=")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{k: v}), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    with open(os.path.join(root, "tests", "golden", "fib582_oracle.json")) as f:
        g = json.load(f)
    assert got["commitments"] == g["commitments"] and got["sha"] == g["proof_sha256"], setting


def test_round_trip_properties_at_scale(prover, rc):
    # size-independent properties at a size the oracle is too slow for in a unit test (cpu 2^16, mem 2^18):
    # the oracle's restated verifier accepts the GPU proof, and a second run reproduces it bit for bit
    w = va.Workload.fib(9359)
    assert w.cpu_height == 1 << 16
    mt, prep = w.main_traces(), w.preprocessed()
    dmain = [prover.upload(m) for m in mt]
    dprep = [(c, prover.upload(m)) for c, m in prep]
    a = prover.prove(dmain, dprep)
    assert po.verify_basic(prep[0][1], prep[1][1], a.words, rc) is None
    assert prover.prove(dmain, dprep).bytes() == a.bytes()
    bad = a.words.copy()
    bad[a.words.size // 2] ^= 1
    assert po.verify_basic(prep[0][1], prep[1][1], bad, rc) is not None


def test_tampered_gpu_proof_rejected(prover, fib25, rc):
    mt = fib25.main_traces()
    prep = fib25.preprocessed()
    proof = prover.prove([prover.upload(m) for m in mt], [(c, prover.upload(m)) for c, m in prep])
    bad = proof.words.copy()
    bad[30] = (int(bad[30]) + 1) % P
    assert po.verify_basic(prep[0][1], prep[1][1], bad, rc) is not None


def test_repeated_proofs_are_identical(prover, fib25):
    mt = fib25.main_traces()
    prep = fib25.preprocessed()
    dmain = [prover.upload(m) for m in mt]
    dprep = [(c, prover.upload(m)) for c, m in prep]
    a = prover.prove(dmain, dprep)
    b = prover.prove(dmain, dprep)
    assert a.bytes() == b.bytes()


# ---- other BASELINE configs as parity cases: C4 (ALU-heavy, multi-chip) and C3's 4x blowup ---------------------
def test_alu_workload_proof_bytes(prover, rc):
    w = va.Workload.alu(100)  # add / sub / bitwise / lt chips populated, range bus busy
    assert w.cpu_height == 1024
    proof, ref, prep = _prove_both(prover, w, rc)
    for chip in range(va.NUM_CHIPS):
        assert first_mismatch(proof.debug_quotient(chip), ref.quotient(chip)) is None, "quotient chunks of chip %d (%s)" % (chip, va.CHIP_NAMES[chip])
    assert first_mismatch(proof.words, ref.words) is None
    assert po.verify_basic(prep[0][1], prep[1][1], proof.words, rc) is None


@pytest.mark.parametrize("make", [lambda: va.Workload.fib(25), lambda: va.Workload.alu(40), lambda: va.Workload.fib(582)])
def test_blowup4_proof_bytes(machine, rc, make):
    p4 = va.Prover(machine, rc, log_blowup=2)
    w = make()
    mt, prep = w.main_traces(), w.preprocessed()
    proof = p4.prove([p4.upload(m) for m in mt], [(c, p4.upload(m)) for c, m in prep])
    ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc, log_blowup=2)
    assert first_mismatch(proof.words, ref.words) is None
    assert po.verify_basic(prep[0][1], prep[1][1], proof.words, rc, log_blowup=2) is None


def test_lde_blowup4_matches_oracle(machine, rc):
    p4 = va.Prover(machine, rc, log_blowup=2)
    rng = np.random.default_rng(77)
    for log_h, w in [(0, 2), (5, 3), (13, 2)]:
        m = rand_matrix(rng, 1 << log_h, w)
        pd = p4.commit_batches([p4.upload(m)])
        assert first_mismatch(pd.lde(0), po.committed_lde(m, 2, 31)) is None
        assert first_mismatch(pd.root, po.commit_root([m], log_blowup=2)) is None


# ---- BASELINE.json's full sizes.  Every proof WORD is compared with the oracle's proof of the same traces (the oracle proves
# a 2^20-row segment in about a minute on the GPU box's host cores), with the committed sha256 fixture of the oracle's proof
# (tests/golden/full_*.json, generated by make_golden.py --full on that host), and through size-independent properties: the
# restated Machine::verify accepts, a second run is identical, tampering is rejected. -------------------------------------
def _golden_full(name):
    import json
    import os

    path = os.path.join(os.path.dirname(__file__), "golden", name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def _full_size_round_trip(p, w, rc, log_blowup, golden=None, oracle_words=True):
    import hashlib

    mt, prep = w.main_traces(), w.preprocessed()
    dmain = [p.upload(m) for m in mt]
    dprep = [(c, p.upload(m)) for c, m in prep]
    a = p.prove(dmain, dprep)
    g = _golden_full(golden) if golden else None
    if g is not None:  # the oracle's proof of this workload, pinned as a fixture
        assert hashlib.sha256(b"".join(m.tobytes() for m in mt)).hexdigest() == g["traces_sha256"]
        assert [int(x) for x in a.words[2:26]] == g["commitments"]
        assert [int(x) for x in a.transcript] == g["transcript"]
        assert int(a.words.size) == g["proof_words"]
        assert hashlib.sha256(a.bytes()).hexdigest() == g["proof_sha256"]
    if oracle_words:   # and word for word against the oracle proving the same traces now
        ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc, log_blowup=log_blowup)
        assert first_mismatch(a.transcript, ref.transcript) is None
        assert first_mismatch(a.words, ref.words) is None
    assert po.verify_basic(prep[0][1], prep[1][1], a.words, rc, log_blowup=log_blowup) is None
    assert p.prove(dmain, dprep).bytes() == a.bytes()  # deterministic: no atomics-order dependence at scale
    for pos in (40, a.words.size // 3, a.words.size - 7):
        bad = a.words.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        assert po.verify_basic(prep[0][1], prep[1][1], bad, rc, log_blowup=log_blowup) is not None


def test_full_size_c2_fib_2_20(prover, rc):
    w = va.Workload.fib(149794)
    assert w.cpu_height == 1 << 20 and w.main_trace(2).shape[0] == 1 << 22
    _full_size_round_trip(prover, w, rc, 1, golden="full_c2_fib149794.json")


def test_full_size_c4_alu_2_20(prover, rc):
    w = va.Workload.alu(116507)
    assert w.cycles == 1048568 and w.cpu_height == 1 << 20
    heights = [m.shape[0] for m in w.main_traces()]
    assert [heights[i] for i in (3, 4, 8, 10)] == [1 << 19, 1 << 17, 1 << 17, 1 << 19]
    # every word is pinned by the fixture (the oracle's proof of this workload, 127 s on the GPU box's 256 host cores)
    _full_size_round_trip(prover, w, rc, 1, golden="full_c4_alu116507.json", oracle_words=False)


def test_full_size_c3_fib_2_22_blowup4(machine, rc):
    w = va.Workload.fib(599183)
    assert w.cpu_height == 1 << 22 and w.main_trace(2).shape[0] == 1 << 24
    # every word is pinned by the fixture (the oracle's proof of this workload: tests/golden/make_golden.py --full .. c3, a quarter of an
    # hour of host time, done once); the same configuration is also compared word for word at 2^18 rows in the next test
    assert _golden_full("full_c3_fib599183_blowup4.json") is not None
    _full_size_round_trip(va.Prover(machine, rc, log_blowup=2), w, rc, 2, golden="full_c3_fib599183_blowup4.json", oracle_words=False)


def test_full_size_c2_poseidon_mmcs(machine, rc, poseidon_oracle):
    # the configuration `bench.py --mmcs poseidon` times (C2 with the Poseidon-16 MMCS): pinned by the oracle's proof of it
    w = va.Workload.fib(149794)
    assert _golden_full("full_c2_fib149794_poseidon.json") is not None
    _full_size_round_trip(va.Prover(machine, rc, hash_kind=va.HASH_POSEIDON16), w, rc, 1, golden="full_c2_fib149794_poseidon.json", oracle_words=False)


def test_full_size_c4_alu_2_20_poseidon_mmcs(machine, rc, poseidon_oracle):
    # BASELINE.json configs[3] to the letter: "alu_u32 + range-check heavy synthetic program, 2^20 rows (multi-chip quotient + Poseidon Merkle)":
    # C4's program committed with the Poseidon-16 MMCS (`bench.py --workload c4 --mmcs poseidon`), every word pinned by the oracle's proof
    w = va.Workload.alu(116507)
    assert _golden_full("full_c4_alu116507_poseidon.json") is not None
    _full_size_round_trip(va.Prover(machine, rc, hash_kind=va.HASH_POSEIDON16), w, rc, 1, golden="full_c4_alu116507_poseidon.json", oracle_words=False)


def test_c3_shape_at_2_18_blowup4_every_word(machine, rc):
    # C3's configuration (4x blowup) at cpu 2^18 / mem 2^20 rows: every proof word against the oracle
    w = va.Workload.fib(37446)
    assert w.cpu_height == 1 << 18 and w.main_trace(2).shape[0] == 1 << 20
    _full_size_round_trip(va.Prover(machine, rc, log_blowup=2), w, rc, 2)


# ---- the two realisations of Chip::eval on the device: ahead-of-time compiled chip kernels (default for the
# in-tree BasicMachine) and the interpreted register program (AIRs captured at run time) -----------------------
@pytest.mark.parametrize("make", [lambda: va.Workload.fib(25), lambda: va.Workload.alu(100)])
def test_interpreted_air_path_matches_oracle(machine, rc, make):
    pi = va.Prover(machine, rc, interpret_air=True)
    w = make()
    proof, ref, prep = _prove_both(pi, w, rc)
    for chip in range(va.NUM_CHIPS):
        assert first_mismatch(proof.debug_quotient(chip), ref.quotient(chip)) is None, "quotient chunks of chip %d (%s)" % (chip, va.CHIP_NAMES[chip])
    assert first_mismatch(proof.words, ref.words) is None


# ---- device trace generation: Chip::generate_trace as kernels, from the VM's operation logs -----
@pytest.mark.parametrize("make", [lambda: va.Workload.fib(25), lambda: va.Workload.alu(100), lambda: va.Workload.fib(582), lambda: va.Workload.alu(1),
                                  lambda: va.Workload.named("static_data"), lambda: va.Workload.named("mixed_ops:40"), lambda: va.Workload.named("mixed_ops:700")])
def test_generated_traces_match_host_generate_trace(prover, make):
    w = make()
    log = prover.upload_oplog(w.oplog())
    for chip in va.GENERATED_CHIPS:
        got = prover.generate_trace(log, chip)
        want = w.main_trace(chip)
        assert got.shape == want.shape, va.CHIP_NAMES[chip]
        assert first_mismatch(got.download(), want) is None, va.CHIP_NAMES[chip]
    with pytest.raises(va.VgpuError):
        prover.generate_trace(log, va.NUM_CHIPS)  # no such chip


def test_proof_from_generated_traces_is_the_same_proof(prover, rc):
    w = va.Workload.alu(100)
    mt, prep = w.main_traces(), w.preprocessed()
    log = prover.upload_oplog(w.oplog())
    dmain = [prover.generate_trace(log, i) if i in va.GENERATED_CHIPS else prover.upload(mt[i]) for i in range(va.NUM_CHIPS)]
    dprep = [(c, prover.upload(m)) for c, m in prep]
    a = prover.prove(dmain, dprep)
    ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc)
    assert first_mismatch(a.words, ref.words) is None


def test_all_fourteen_generators_feed_a_proof_that_is_the_oracles(prover, rc):
    """mixed_ops keeps EVERY chip busy (mul / mulhs / mulhu, div / sdiv, shl / shr / sra, ne / eq, write beside add, sub, xor, lt): all fourteen
    traces generated on the device from the operation logs, and the proof of them word for word the oracle's proof of the host-generated traces.
    (The reference's mul / div / shift / com chips are stubs, SURVEY.md 0.6: such a proof is well-formed but no verifier accepts it — in the
    reference either; what is pinned here is that the device reproduces the reference's generate_trace for them, incomplete as it is.)"""
    w = va.Workload.named("mixed_ops:300")
    mt, prep = w.main_traces(), w.preprocessed()
    assert [m.shape[0] for m in mt][5:8] == [2048, 2048, 2048] and mt[9].shape[0] == 1024 and mt[11].shape[0] == 512  # mul, div, shift, com, output all busy
    log = prover.upload_oplog(w.oplog())
    dmain = [prover.generate_trace(log, i) for i in range(va.NUM_CHIPS)]
    a = prover.prove(dmain, [(c, prover.upload(m)) for c, m in prep])
    ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc)
    assert first_mismatch(a.words, ref.words) is None
    assert po.verify_basic(prep[0][1], prep[1][1], a.words, rc) is not None  # stub chips: rejected, as the reference's own verifier would


def test_operation_logs_are_validated_at_upload(prover):
    import ctypes

    w = va.Workload.named("mixed_ops:8")
    d = w.oplog()
    n = int(d.n_alu2[2])
    ops = np.ctypeslib.as_array(ctypes.cast(d.alu2[2], ctypes.POINTER(ctypes.c_uint32)), shape=(n, 4)).copy()
    bad = va.OplogDesc()
    ctypes.memmove(ctypes.byref(bad), ctypes.byref(d), ctypes.sizeof(d))
    ops[3, 0] = 100  # an ADD32 in the shift chip's log
    bad.alu2[2] = ops.ctypes.data
    with pytest.raises(va.VgpuError, match="not an operation of that chip"):
        prover.upload_oplog(bad)
    ops[3, 0], ops[3, 3] = 105, 40  # SHL32 by 40
    with pytest.raises(va.VgpuError, match="shift amount"):
        prover.upload_oplog(bad)
    cpu = np.ctypeslib.as_array(ctypes.cast(d.cpu, ctypes.POINTER(ctypes.c_uint32)), shape=(int(d.n_cpu), 12)).copy()
    bus = int(np.nonzero(cpu[:, 8] == 7)[0][0])  # first bus operation
    cpu[bus, 2] = 9  # READ_ADVICE: no chip of this machine receives it
    bad2 = va.OplogDesc()
    ctypes.memmove(ctypes.byref(bad2), ctypes.byref(d), ctypes.sizeof(d))
    bad2.cpu = cpu.ctypes.data
    with pytest.raises(va.VgpuError, match="no bus operation"):
        prover.upload_oplog(bad2)


def test_generated_traces_full_size(prover, rc):
    # C2 at full size: memory log of 2^22 - O(1) entries through the device sort; proof identical to the uploaded-trace proof
    w = va.Workload.fib(149794)
    mt, prep = w.main_traces(), w.preprocessed()
    log = prover.upload_oplog(w.oplog())
    gen = {i: prover.generate_trace(log, i) for i in va.GENERATED_CHIPS}
    for i in (0, 2, 3):
        assert first_mismatch(gen[i].download(), mt[i]) is None, va.CHIP_NAMES[i]
    dprep = [(c, prover.upload(m)) for c, m in prep]
    a = prover.prove([gen[i] if i in gen else prover.upload(mt[i]) for i in range(va.NUM_CHIPS)], dprep)
    b = prover.prove([prover.upload(m) for m in mt], dprep)
    assert a.bytes() == b.bytes()


# ---- the reference's other pinned prover programs (basic/tests/test_prover.rs:190-402, asserted :489-640) ---------
@pytest.mark.parametrize("name", ["left_imm_ops", "signed_inequality", "loadfp", "static_data"])
@pytest.mark.parametrize("interpret", [False, True])
def test_reference_test_programs_proof_bytes(machine, rc, name, interpret):
    p = va.Prover(machine, rc, interpret_air=interpret)
    w = va.Workload.named(name)
    proof, ref, prep = _prove_both(p, w, rc)
    for chip in range(va.NUM_CHIPS):
        assert first_mismatch(proof.debug_perm_trace(chip), ref.perm_trace(chip)) is None, "perm trace of chip %d" % chip
        assert first_mismatch(proof.debug_quotient(chip), ref.quotient(chip)) is None, "quotient chunks of chip %d (%s)" % (chip, va.CHIP_NAMES[chip])
    assert first_mismatch(proof.words, ref.words) is None
    assert po.verify_basic(prep[0][1], prep[1][1], proof.words, rc) is None
    log = p.upload_oplog(w.oplog())
    for chip in va.GENERATED_CHIPS:
        assert first_mismatch(p.generate_trace(log, chip).download(), w.main_trace(chip)) is None, va.CHIP_NAMES[chip]


# ---- error behaviour at the boundary: the reference panics (basic/src/lib.rs:210,624,643-645); the C ABI returns a
# status and a message instead, never unwinds, and stays usable afterwards --------------------------------------
def test_bad_inputs_are_rejected_not_crashed(prover, fib25, rc):
    mt, prep = fib25.main_traces(), fib25.preprocessed()
    dmain = [prover.upload(m) for m in mt]
    dprep = [(c, prover.upload(m)) for c, m in prep]
    with pytest.raises(va.VgpuError, match="one main trace per chip"):
        prover.prove(dmain[:-1], dprep)
    with pytest.raises(va.VgpuError, match="width mismatch"):
        prover.prove([dmain[1]] + dmain[1:], dprep)
    with pytest.raises(va.VgpuError, match="powers of two"):
        prover.prove([prover.upload(mt[0][:192])] + dmain[1:], dprep)
    with pytest.raises(va.VgpuError, match="shape mismatch"):
        prover.prove(dmain, [(dprep[0][0], prover.upload(prep[0][1][:16])), dprep[1]])
    with pytest.raises(va.VgpuError):
        prover.commit_batches([prover.upload(mt[0][:100])])  # height not a power of two
    with pytest.raises(va.VgpuError):
        prover.fri_fold(np.zeros((6, 5), dtype=np.uint32), np.zeros(5, dtype=np.uint32))
    # the context is still good: the same prover produces the reference proof afterwards
    ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc)
    assert first_mismatch(prover.prove(dmain, dprep).words, ref.words) is None


# ---- LDE at the large heights: both four-step splits (2^12-point tiles up to 2^22, 2^14-point tiles from 2^23) ------
@pytest.mark.parametrize("log_h,w,log_blowup", [(13, 5, 1), (14, 3, 2), (16, 2, 1), (18, 3, 1), (20, 2, 1), (22, 1, 1), (23, 1, 1), (24, 1, 1), (22, 1, 2), (23, 2, 2),
                                                   (24, 1, 2)])  # (24, 1, 2): the 2^26-row LDE of C3's memory chip, element for element
def test_large_lde_matches_oracle(machine, rc, log_h, w, log_blowup):
    p = va.Prover(machine, rc, log_blowup=log_blowup)
    rng = np.random.default_rng(4000 + log_h)
    m = rand_matrix(rng, 1 << log_h, w)
    pd = p.commit_batches([p.upload(m)])
    assert first_mismatch(pd.lde(0), po.committed_lde(m, log_blowup, 31)) is None


# ---- the foreign-host route: chips captured through the vgpu_air_* FFI, proved by the interpreted programs -----------
@pytest.mark.parametrize("make", [lambda: va.Workload.fib(25), lambda: va.Workload.alu(100), lambda: va.Workload.named("signed_inequality")])
def test_ffi_captured_machine_proof_bytes(rc, make):
    p = va.Prover(va.Machine.basic_via_ffi(), rc)
    w = make()
    proof, ref, prep = _prove_both(p, w, rc)
    assert first_mismatch(proof.words, ref.words) is None
    with pytest.raises(va.VgpuError):
        p.generate_trace(p.upload_oplog(w.oplog()), 0)  # device trace generators belong to the in-tree chips


# ---- FriConfig variants (basic/tests/test_prover.rs:441-446) and the final-poly convention switch ------------------
@pytest.mark.parametrize("num_queries,pow_bits,observe", [(1, 0, False), (7, 3, False), (40, 12, False), (40, 8, True), (100, 1, True)])
def test_fri_config_variants_proof_bytes(machine, rc, fib25, num_queries, pow_bits, observe):
    p = va.Prover(machine, rc, num_queries=num_queries, pow_bits=pow_bits, observe_final_poly=observe)
    mt, prep = fib25.main_traces(), fib25.preprocessed()
    proof = p.prove([p.upload(m) for m in mt], [(c, p.upload(m)) for c, m in prep])
    po.set_observe_final_poly(observe)
    try:
        ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc, num_queries=num_queries, pow_bits=pow_bits)
        assert first_mismatch(proof.words, ref.words) is None
        assert po.verify_basic(prep[0][1], prep[1][1], proof.words, rc, num_queries=num_queries, pow_bits=pow_bits) is None
    finally:
        po.set_observe_final_poly(False)


def test_other_poseidon_constants(machine, fib25):
    # the reference draws the 480 round constants from thread_rng (basic/tests/test_prover.rs:418-422): any set must work
    rng = np.random.default_rng(2024)
    rc2 = rng.integers(0, P, size=480, dtype=np.uint32)
    p = va.Prover(machine, rc2)
    mt, prep = fib25.main_traces(), fib25.preprocessed()
    proof = p.prove([p.upload(m) for m in mt], [(c, p.upload(m)) for c, m in prep])
    ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc2)
    assert first_mismatch(proof.words, ref.words) is None
    assert po.verify_basic(prep[0][1], prep[1][1], proof.words, rc2) is None
    assert po.verify_basic(prep[0][1], prep[1][1], proof.words, va.poseidon_round_constants()) is not None


def test_two_provers_in_flight_produce_the_reference_proofs(machine, rc):
    # bench.py keeps two proofs in flight per GPU from two host threads, each with its own prover context: the
    # contexts must not disturb each other (every proof still the oracle's proof, bit for bit)
    import threading

    works = [va.Workload.fib(582), va.Workload.alu(300)]
    refs = []
    for w in works:
        prep = w.preprocessed()
        refs.append(po.prove_basic(w.main_traces(), prep[0][1], prep[1][1], rc).words)
    provers = [va.Prover(machine, rc) for _ in works]
    results = [[] for _ in works]

    def worker(i):
        p, w = provers[i], works[i]
        mt, prep = w.main_traces(), w.preprocessed()
        dmain = [p.upload(m) for m in mt]
        dprep = [(c, p.upload(m)) for c, m in prep]
        for _ in range(6):
            results[i].append(p.prove(dmain, dprep).words)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(works))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(len(works)):
        assert len(results[i]) == 6
        for got in results[i]:
            assert first_mismatch(got, refs[i]) is None


# ---- invalid witnesses: the device prover never "fixes" them — the proof is rejected (or proving fails loudly) ----------
def test_invalid_traces_do_not_yield_accepted_proofs(prover, fib25, rc):
    mt, prep = fib25.main_traces(), fib25.preprocessed()
    dprep = [(c, prover.upload(m)) for c, m in prep]

    def outcome(traces):
        try:
            proof = prover.prove([prover.upload(m) for m in traces], dprep)
        except va.VgpuError as e:
            return "error: " + str(e)
        return po.verify_basic(prep[0][1], prep[1][1], proof.words, rc)

    assert outcome(mt) is None
    # (1) a constraint violation: one output byte of an ADD32 row is wrong -> the quotient is not a polynomial
    bad = [m.copy() for m in mt]
    bad[3][5, 11] = (int(bad[3][5, 11]) + 1) % P
    assert outcome(bad) is not None
    # (2) constraints hold but a bus is unbalanced: one range-check multiplicity too many -> cumulative sums do not cancel
    bad = [m.copy() for m in mt]
    bad[12][7, 0] = (int(bad[12][7, 0]) + 1) % P
    assert "cumulative" in (outcome(bad) or "").lower() or outcome(bad) is not None
    # (3) the cpu trace claims another opcode flag on one row
    bad = [m.copy() for m in mt]
    row = int(np.nonzero(bad[0][:, 7] == 1)[0][0]) if (bad[0][:, 7] == 1).any() else 3
    bad[0][row, 7] ^= 1
    assert outcome(bad) is not None


def test_async_prove_matches_sync_prove(machine, rc, fib25):
    # vgpu_prove_async / vgpu_ticket_wait: two contexts, one ticket outstanding on each, single caller thread
    provers = [va.Prover(machine, rc) for _ in range(2)]
    mt, prep = fib25.main_traces(), fib25.preprocessed()
    ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc).words
    inputs = [([p.upload(m) for m in mt], [(c, p.upload(m)) for c, m in prep]) for p in provers]
    tickets = [p.prove_async(*inp) for p, inp in zip(provers, inputs)]
    for t in tickets:
        assert first_mismatch(t.wait().words, ref) is None
    # an error in the worker surfaces at wait() with its message
    bad = provers[0].prove_async(inputs[0][0][:-1], inputs[0][1])
    with pytest.raises(va.VgpuError, match="one main trace per chip"):
        bad.wait()
    assert first_mismatch(provers[0].prove(*inputs[0]).words, ref) is None


# ---- check_constraints / check_cumulative_sums on the device (debug builds of the reference, basic/src/lib.rs:270-375) ----
@pytest.mark.parametrize("interpret", [False, True])
def test_device_check_constraints(machine, rc, fib25, interpret):
    p = va.Prover(machine, rc, interpret_air=interpret)
    mt, prep = fib25.main_traces(), fib25.preprocessed()
    dprep = [(c, p.upload(m)) for c, m in prep]
    ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc).words
    ok = p.prove([p.upload(m) for m in mt], dprep, check=True)  # a valid witness passes and the proof is unchanged
    assert first_mismatch(ok.words, ref) is None
    for w in (va.Workload.alu(50), va.Workload.named("signed_inequality"), va.Workload.named("static_data")):
        wp = w.preprocessed()
        p.prove([p.upload(m) for m in w.main_traces()], [(c, p.upload(m)) for c, m in wp], check=True)

    def failure(traces):
        with pytest.raises(va.VgpuError) as e:
            p.prove([p.upload(m) for m in traces], dprep, check=True)
        return str(e.value)

    bad = [m.copy() for m in mt]
    bad[3][5, 11] = (int(bad[3][5, 11]) + 1) % P  # ADD32 row 5: output byte 0 (column 11) is off by one
    msg = failure(bad)
    assert "chip add" in msg.lower() and "row 5" in msg and ("AIR constraint 3" in msg if not interpret else "AIR constraint" in msg)
    bad = [m.copy() for m in mt]
    bad[12][7, 0] = (int(bad[12][7, 0]) + 1) % P  # one range multiplicity too many: every chip is fine, the bus is not
    assert "cumulative sums" in failure(bad)
    bad = [m.copy() for m in mt]
    bad[0][100, 1] = (int(bad[0][100, 1]) + 4) % P  # the cpu's frame pointer jumps on row 100
    msg = failure(bad)
    assert "chip cpu" in msg.lower() and ("row 99" in msg or "row 100" in msg)


def test_handles_may_be_freed_in_any_order(machine, rc, fib25):
    # a garbage collector frees handles in arbitrary order: device objects keep their prover's context and pool alive
    import gc

    p = va.Prover(machine, rc)
    t = p.upload(fib25.main_trace(0))
    pd = p.commit_batches([t])
    log = p.upload_oplog(fib25.oplog())
    g = p.generate_trace(log, 0)
    # drop the Python-side back references, then free the prover FIRST
    for obj in (t, pd, log, g):
        obj._prover = None
    h = p._h
    p._h = None
    va.lib().vgpu_prover_destroy(h)
    gc.collect()
    del t, pd, log, g  # the buffers return to a pool that must still exist
    gc.collect()
    q = va.Prover(machine, rc)  # and the device is still usable
    assert q.upload(fib25.main_trace(3)).shape == fib25.main_trace(3).shape


def test_pool_trim_returns_cached_blocks(machine, rc, fib25):
    p = va.Prover(machine, rc)
    mt, prep = fib25.main_traces(), fib25.preprocessed()
    dmain = [p.upload(m) for m in mt]
    dprep = [(c, p.upload(m)) for c, m in prep]
    a = p.prove(dmain, dprep)
    live_before, _ = p.memory()
    freed = p.trim()
    assert freed > 0 and p.memory()[0] == live_before  # live blocks (the uploaded traces) untouched
    assert p.prove(dmain, dprep).bytes() == a.bytes()     # and the prover simply allocates again


def test_lde_height_beyond_two_adicity_is_rejected(machine, rc):
    # BabyBear has no multiplicative subgroup of order 2^28: the reference would panic inside two_adic_generator
    p = va.Prover(machine, rc, log_blowup=2)
    t = p.upload(np.zeros((1 << 26, 1), dtype=np.uint32))
    with pytest.raises(va.VgpuError, match="two-adicity"):
        p.commit_batches([t])
    small = np.arange(64, dtype=np.uint32).reshape(16, 4) % P
    assert first_mismatch(p.commit_batches([p.upload(small)]).root, po.commit_root([small], log_blowup=2)) is None


@pytest.mark.parametrize("make", [lambda: va.Workload.fib(25), lambda: va.Workload.alu(40)])
def test_blowup8_proof_bytes(machine, rc, make):
    # FriConfig.log_blowup = 3: the quotient domain is a strided subset of the LDE (machine/src/quotient.rs:41-47)
    p8 = va.Prover(machine, rc, log_blowup=3)
    w = make()
    mt, prep = w.main_traces(), w.preprocessed()
    proof = p8.prove([p8.upload(m) for m in mt], [(c, p8.upload(m)) for c, m in prep])
    ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc, log_blowup=3)
    assert first_mismatch(proof.words, ref.words) is None
    assert po.verify_basic(prep[0][1], prep[1][1], proof.words, rc, log_blowup=3) is None


@pytest.mark.parametrize("launcher", ["self-spawn", "torchrun"])
def test_bench_two_rank_control_flow_on_one_gpu(launcher):
    # the N > 1 path of bench.py end to end (prover contexts per rank, barriers, the batched root all-gather, MAX over ranks,
    # one JSON line from rank 0) with both ranks on this GPU and gloo standing in for RCCL.  "self-spawn": `python bench.py --gpus 2` exactly
    # as the N = 1 line is started (round-5 verdict, item 1: that form used to exit with status 2); "torchrun": the driver's documented form.
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VGPU_BENCH_BACKEND="gloo", VGPU_BENCH_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    port = 29600 + os.getpid() % 300
    head = [sys.executable] if launcher == "self-spawn" else [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    out = subprocess.run(head + [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--log-rows", "14", "--no-cpu-baseline",
                                 "--no-extra-legs", "--sustained-seconds", "0.5"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    c = d["config"]["collective"]
    assert c["backend"] == "gloo" and c["ranks_in_last_allgather"] == 2 and c["distinct_root_sets"] == 2  # both ranks' roots arrived, and they are different segments
    assert [r["rank"] for r in c["rank_devices"]] == [0, 1] and c["distinct_devices"] == 1  # the one-device stand-in says so
    sus = d["sustained"]
    assert sus["seconds"] >= 0.2 and sus["steps"] >= 6  # (the region's length is a step COUNT sized from the contract region's rate; at this toy size the first steps are the slow ones) and abs(sus["proofs_per_s"] - 2 * sus["steps"] / sus["seconds"]) < 1e-6 * sus["proofs_per_s"]


def test_bench_self_spawn_returns_a_failing_ranks_status():
    # a rank that dies takes the launcher down with its status instead of leaving the other rank at a barrier until the driver's limit
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VGPU_BENCH_BACKEND="gloo", VGPU_BENCH_DEVICE="0", VGPU_BENCH_TEST_EXIT_RANK="1")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-rows", "12", "--no-cpu-baseline",
                          "--no-extra-legs", "--sustained-seconds", "0"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 7 and "rank 1 ended with status 7" in out.stderr, (out.returncode, out.stderr[-800:])
    assert time.time() - t0 < 120 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


# ---- the fine-grained PCS / quotient boundary: a host that drives the phases of Machine::prove itself through
# commit_batches / get_ldes / generate_permutation_trace / quotient / open_multi_batches + its own challenger gets the same proof
def _ext_mul_base(e, b):
    return [int(po.lib().oracle_fp_mul(ctypes.c_uint32(int(x)), ctypes.c_uint32(int(b)))) for x in e]


import ctypes  # noqa: E402


@pytest.mark.parametrize("make", [lambda: va.Workload.fib(25), lambda: va.Workload.alu(40), lambda: va.Workload.named("static_data")])
def test_proof_assembled_from_fine_grained_calls_is_the_same_proof(machine, rc, make):
    p = va.Prover(machine, rc)
    w = make()
    mt, prep = w.main_traces(), w.preprocessed()
    dmain = [p.upload(m) for m in mt]
    dprep = [(c, p.upload(m)) for c, m in prep]
    whole = p.prove(dmain, dprep)

    ch = va.Challenger(rc)                                              # basic/src/lib.rs:185
    prep_pd = p.commit_batches([t for _, t in dprep])                   # :199
    ch.observe(prep_pd.root)
    main_pd = p.commit_batches(dmain)                                   # :223
    ch.observe(main_pd.root)
    rnd = ch.sample(15)                                                 # :227-230
    prep_slot = {c: k for k, (c, _) in enumerate(dprep)}
    perm, cums = [], []
    for i in range(14):                                                 # :232-250
        t, cs = p.permutation_trace_device(i, dmain[i], rnd, dprep[prep_slot[i]][1] if i in prep_slot else None)
        perm.append(t)
        cums.append(cs)
    perm_pd = p.commit_batches(perm)                                    # :258
    ch.observe(perm_pd.root)
    alpha = ch.sample(5)                                                # :263
    v = main_pd.lde_view(0)                                             # get_ldes (:225): a device view, no copy
    assert v["data"] and (v["height"], v["width"], v["stride"], v["log_blowup"]) == (mt[0].shape[0] * 2, mt[0].shape[1], mt[0].shape[0] * 2, 1)
    assert (main_pd.num_matrices, perm_pd.num_matrices, prep_pd.num_matrices) == (14, 14, 2)
    quot = [p.quotient(i, main_pd, i, perm_pd, i, rnd, alpha, cums[i], prep_pd if i in prep_slot else None, prep_slot.get(i, 0)) for i in range(14)]  # :284-590
    assert all(q.shape == (mt[i].shape[0], 10) for i, q in enumerate(quot))
    quot_pd = p.commit_batches(quot, coset_shifts=[31 * 31] * 14)       # :593-599: coset_shift^(2^log_quotient_degree)
    ch.observe(quot_pd.root)
    zeta = [int(x) for x in ch.sample(5)]                               # :606
    zeta2 = [int(x) for x in po.ext5_mul(zeta, zeta)]
    points = [[], [], []]
    for i in range(14):
        g = int(po.lib().oracle_two_adic_generator(ctypes.c_uint32(int(np.log2(mt[i].shape[0])))))
        points[0].append([zeta, _ext_mul_base(zeta, g)])
        points[1].append([zeta, _ext_mul_base(zeta, g)])
        points[2].append([zeta2])
    opened, pcs_words = p.open_multi_batches([main_pd, perm_pd, quot_pd], points, ch)   # :611-619

    words = [0x31465056, 14] + [int(x) for x in main_pd.root] + [int(x) for x in perm_pd.root] + [int(x) for x in quot_pd.root]
    for i in range(14):                                                 # MachineProof / ChipProof (machine/src/proof.rs:13-44)
        words.append(int(np.log2(mt[i].shape[0])))
        for vec in (opened[0][i][0], opened[0][i][1], opened[1][i][0], opened[1][i][1], opened[2][i][0]):
            words.append(vec.shape[0])
            words += [int(x) for x in vec.ravel()]
        words += [int(x) for x in cums[i]]
    words += [int(x) for x in pcs_words]
    assert first_mismatch(np.array(words, dtype=np.uint32), whole.words) is None
    # the quotient chunks are the ones the whole prover computed (natural row order after undoing the bit reversal)
    dbg = p.prove(dmain, dprep, debug=True)
    for i in (0, 3, 12):
        q = quot[i].download()
        n = q.shape[0]
        k = int(np.log2(n))
        nat = np.array([int(format(r, "0%db" % k)[::-1], 2) if k else 0 for r in range(n)])
        assert first_mismatch(q[nat], dbg.debug_quotient(i)) is None


@pytest.mark.parametrize("shapes", [
    [[(64, 130), (64, 3)], [(256, 2), (8, 7), (1, 4)]],                       # every matrix below the matrix-core threshold (VALU k_col_dot)
    [[(2048, 130), (1024, 3)], [(4096, 17), (1024, 33), (64, 5)]],            # k_col_dot_mfma: column groups of 16 with ragged tails, chunked columns
    # trees of >= 2^16 leaves, whose bottom digest layers are NOT kept (round 6, DeviceTree::drop_bottom): round 0 injects a matrix at layer 1 (only the leaf
    # layer goes; its siblings are recomputed as row hashes), round 1 drops two layers (level-1 siblings = compress of two recomputed row hashes); the FRI
    # layer trees of 2^16 pairs likewise
    [[(1 << 15, 3), (1 << 14, 2)], [(1 << 16, 2), (1 << 13, 4), (1 << 16, 1)]],
])
def test_open_multi_batches_generic_shapes_match_oracle(machine, rc, shapes):
    """Shapes Machine::prove never produces: one round with a 130-column matrix (column chunks in k_col_dot), five points on one
    matrix (point chunks, > 4 distinct points on one LDE height), a second round with mixed heights — against the oracle's
    pcs.commit_batches + open_multi_batches on the same matrices, points and transcript prefix."""
    p = va.Prover(machine, rc, num_queries=9, pow_bits=3)
    rng = np.random.default_rng(99)
    rounds = [[rand_matrix(rng, h, w) for h, w in rnd] for rnd in shapes]
    ext = lambda: [int(x) for x in rng.integers(1, P, 5)]
    pts = [ext() for _ in range(6)]
    points = [[[pts[0]], [pts[0], pts[1], pts[2], pts[3], pts[4]]], [[pts[5], pts[0]], [pts[1]], [pts[2], pts[3], pts[4]]]]
    if rounds[0][0].shape[0] >= 1 << 15:  # the tall case: fewer points (the oracle's barycentric sums are the slow part there)
        points = [[[pts[0]], [pts[0], pts[1]]], [[pts[5], pts[0]], [pts[1]], [pts[2]]]]
    obs = [int(x) for x in rng.integers(0, P, 11)]
    roots, values, proof = po.pcs_open(rounds, points, rc, observed=obs, num_queries=9, pow_bits=3)
    pds = [p.commit_batches([p.upload(m) for m in rnd]) for rnd in rounds]
    for k, pd in enumerate(pds):
        assert first_mismatch(pd.root, roots[k]) is None
    ch = va.Challenger(rc)
    ch.observe(obs)
    opened, words = p.open_multi_batches(pds, points, ch)
    got = np.concatenate([v.ravel() for rnd in opened for mat in rnd for v in mat])
    assert first_mismatch(got, values) is None
    assert first_mismatch(words, proof) is None
    # pcs.verify_multi_batches of the product (host side) accepts what the device produced, from the same transcript prefix
    ch2 = va.Challenger(rc)
    ch2.observe(obs)
    heights, widths = [[m.shape[0] for m in rnd] for rnd in rounds], [[m.shape[1] for m in rnd] for rnd in rounds]
    assert va.verify_multi_batches([pd.root for pd in pds], heights, widths, points, opened, words, ch2, rc, num_queries=9, pow_bits=3) is None
    # and both transcripts were advanced identically by open / verify: their next samples agree
    assert np.array_equal(ch.sample(7), ch2.sample(7))


# ---- Poseidon-16 MMCS (vgpu_config.hash_kind = VGPU_HASH_POSEIDON16): BASELINE.json's north-star Merkle variant ---------------
@pytest.fixture
def poseidon_oracle(rc):
    po.set_mmcs_hash(1, rc)
    yield
    po.set_mmcs_hash(0)


def test_poseidon_mmcs_commit_roots(machine, rc, poseidon_oracle):
    p = va.Prover(machine, rc, hash_kind=va.HASH_POSEIDON16)
    rng = np.random.default_rng(5)
    # widths around the sponge rate (8), mixed heights with injections at several layers, a one-row matrix, a tall one (2^13 rows:
    # separate compress launches above the 1024-parent top kernel)
    batches = [[rand_matrix(rng, 16, 7)], [rand_matrix(rng, 16, 8)], [rand_matrix(rng, 16, 9)], [rand_matrix(rng, 1, 3)],
               [rand_matrix(rng, 64, 17), rand_matrix(rng, 64, 5), rand_matrix(rng, 16, 30), rand_matrix(rng, 2, 4), rand_matrix(rng, 1, 11)],
               [rand_matrix(rng, 1 << 13, 3), rand_matrix(rng, 1 << 11, 20)]]
    for mats in batches:
        pd = p.commit_batches([p.upload(m) for m in mats])
        assert first_mismatch(pd.root, po.commit_root(mats)) is None, [m.shape for m in mats]


def test_poseidon_open_of_trees_that_dropped_their_bottom_layers(machine, rc, poseidon_oracle):
    """As the tall case of test_open_multi_batches_generic_shapes_match_oracle with the Poseidon-16 MMCS: the sibling digests of the dropped layers come from
    k_poseidon_bottom_q (row permutation, 32 lanes per sibling) — one round with an injection at layer 1 (one layer dropped), one without (two)."""
    p = va.Prover(machine, rc, num_queries=9, pow_bits=3, hash_kind=va.HASH_POSEIDON16)
    rng = np.random.default_rng(77)
    rounds = [[rand_matrix(rng, 1 << 15, 9), rand_matrix(rng, 1 << 14, 2)], [rand_matrix(rng, 1 << 16, 2), rand_matrix(rng, 1 << 13, 17)]]
    ext = lambda: [int(x) for x in rng.integers(1, P, 5)]
    pts = [ext() for _ in range(3)]
    points = [[[pts[0]], [pts[0], pts[1]]], [[pts[2], pts[0]], [pts[1]]]]
    obs = [int(x) for x in rng.integers(0, P, 5)]
    roots, values, proof = po.pcs_open(rounds, points, rc, observed=obs, num_queries=9, pow_bits=3)
    pds = [p.commit_batches([p.upload(m) for m in rnd]) for rnd in rounds]
    for k, pd in enumerate(pds):
        assert first_mismatch(pd.root, roots[k]) is None
    ch = va.Challenger(rc)
    ch.observe(obs)
    opened, words = p.open_multi_batches(pds, points, ch)
    assert first_mismatch(np.concatenate([v.ravel() for rnd in opened for mat in rnd for v in mat]), values) is None
    assert first_mismatch(words, proof) is None


@pytest.mark.parametrize("make", [lambda: va.Workload.fib(25), lambda: va.Workload.alu(40), lambda: va.Workload.fib(582)])
def test_poseidon_mmcs_proof_bytes(machine, rc, poseidon_oracle, make):
    p = va.Prover(machine, rc, hash_kind=va.HASH_POSEIDON16)
    w = make()
    mt, prep = w.main_traces(), w.preprocessed()
    proof = p.prove([p.upload(m) for m in mt], [(c, p.upload(m)) for c, m in prep])
    ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc)
    assert first_mismatch(proof.transcript, ref.transcript) is None
    assert first_mismatch(proof.words, ref.words) is None
    assert po.verify_basic(prep[0][1], prep[1][1], proof.words, rc) is None
    po.set_mmcs_hash(0)  # the Keccak verifier must not accept a Poseidon-committed proof
    assert po.verify_basic(prep[0][1], prep[1][1], proof.words, rc) is not None


# ---- §8(f)-4: one commitment round of one proof sharded over W ranks, and RCCL inside the library --------------------------
@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("hash_kind", [va.HASH_KECCAK256, va.HASH_POSEIDON16])
def test_sharded_commit_gives_the_single_gpu_root(machine, rc, world, hash_kind):
    """Column-sharded LDE -> all-to-all into row ranges -> per-rank subtrees -> gathered roots -> top: `world` prover contexts on this
    one GPU stand in for the ranks (the exchanges are device-to-device copies; over RCCL the same phases run in
    vgpu_commit_batches_sharded).  Mixed heights, matrices shorter than the rank count, column counts that do not divide."""
    provers = [va.Prover(machine, rc, hash_kind=hash_kind) for _ in range(world)]
    rng = np.random.default_rng(world)
    mats = [rand_matrix(rng, 1 << 10, 7), rand_matrix(rng, 1 << 12, 3), rand_matrix(rng, 1 << 10, 1), rand_matrix(rng, 64, 19), rand_matrix(rng, 2, 5),
            rand_matrix(rng, 1, 4), rand_matrix(rng, 1, 2)]
    want = provers[0].commit_batches([provers[0].upload(m) for m in mats]).root
    assert first_mismatch(va.commit_batches_sharded_local(provers, mats), want) is None
    shifts = [int(x) for x in rng.integers(1, P, len(mats))]
    want = provers[0].commit_batches([provers[0].upload(m) for m in mats], coset_shifts=shifts).root
    assert first_mismatch(va.commit_batches_sharded_local(provers, mats, coset_shifts=shifts), want) is None


def test_sharded_commit_of_the_basic_machine_main_round(machine, rc):
    w = va.Workload.fib(582)
    mt = w.main_traces()
    provers = [va.Prover(machine, rc) for _ in range(4)]
    want = provers[0].commit_batches([provers[0].upload(m) for m in mt]).root
    assert first_mismatch(va.commit_batches_sharded_local(provers, mt), want) is None


def test_rccl_communicator_of_the_library_world_of_one(prover, fib25):
    """RCCL is loaded and driven by the library itself (vgpu_comm_*): a world of one rank on this box's single GPU exercises the
    load, the communicator and both collectives' call paths (two ranks cannot share one device under RCCL)."""
    comm = va.Comm(prover, va.Comm.unique_id(), 0, 1)
    roots = np.arange(24, dtype=np.uint32)
    assert np.array_equal(comm.allgather_roots(roots), roots.reshape(1, 24))
    mt = fib25.main_traces()
    dm = [prover.upload(m) for m in mt]
    assert first_mismatch(comm.commit_batches_sharded(dm), prover.commit_batches(dm).root) is None
    # the same collectives under a deadline (vgpu_comm_set_timeout_ms: event + ncclCommGetAsyncError polling instead of a stream synchronisation;
    # what bench.py --gpus N runs with), and a whole sharded proof through the deadline-bounded exchanges of the RCCL fabric
    comm.set_timeout_ms(20000)
    assert np.array_equal(comm.allgather_roots(roots[::-1].copy()), roots[::-1].reshape(1, 24))
    prep = fib25.preprocessed()
    single = prover.prove(dm, [(c, prover.upload(m)) for c, m in prep])
    sharded = comm.prove_sharded([prover.upload(m) for m in mt], [(c, prover.upload(m)) for c, m in prep], log_min_sharded=4)
    assert first_mismatch(sharded.words, single.words) is None


@pytest.mark.gpu
@pytest.mark.parametrize("specs,heights,log_blowup", [
    ([("pow5", 5, False)], [16], 2),
    ([("pow5", 5, False)], [1 << 12], 3),
    ([("pow9", 9, True), ("pow5", 5, False)], [8, 32], 3),
    ([("pow9", 9, True)], [1 << 14], 3),
    ([("pow5", 5, False), ("pow9", 9, True), ("pow5", 5, False)], [1 << 10, 2, 1 << 13], 3),
])
def test_general_log_quotient_degree_proof_bytes(rc, specs, heights, log_blowup):
    """AIRs of degree 5 / 9 (log_quotient_degree 2 / 3; machine/src/quotient.rs handles any): the quotient is evaluated on
    2^lqd cosets, decomposed into 2^lqd chunks, committed and opened at zeta^(2^lqd) -- every proof word equals the oracle's."""
    from conftest import pow_machine, pow_trace

    mach, codes = pow_machine(specs)
    assert codes == [0] * len(specs)
    ids = [po.TEST_POW9 if d == 9 else po.TEST_POW5 for _, d, _ in specs]
    traces = [pow_trace(h, d, 3) for h, (_, d, _) in zip(heights, specs)]
    p = va.Prover(mach, rc, log_blowup=log_blowup, num_queries=11, pow_bits=4)
    proof = p.prove([p.upload(t) for t in traces], [], check=True)
    ref = po.prove_machine(ids, traces, rc, log_blowup=log_blowup, num_queries=11, pow_bits=4)
    assert first_mismatch(proof.words, ref.words) is None
    assert po.verify_machine(ids, proof.words, rc, log_blowup=log_blowup, num_queries=11, pow_bits=4) is None
    bad = traces[0].copy()
    bad[1, 1] ^= 1  # y != x^d on one row: the proof must not verify
    traces[0] = bad
    proof = p.prove([p.upload(t) for t in traces], [])
    assert po.verify_machine(ids, proof.words, rc, log_blowup=log_blowup, num_queries=11, pow_bits=4) is not None


@pytest.mark.gpu
def test_quotient_degree_above_the_blowup_is_refused(rc):
    from conftest import pow_machine

    mach, _ = pow_machine([("pow5", 5, False)])
    with pytest.raises(va.VgpuError) as e:
        va.Prover(mach, rc, log_blowup=1)
    assert "log_blowup" in str(e.value)


@pytest.mark.gpu
def test_upload_from_page_locked_host_memory(prover, fib25):
    """vgpu_host_alloc: matrices handed over in page-locked memory give the same proof; the allocation outlives views of it."""
    import gc

    ref = prover.prove([prover.upload(fib25.main_trace(c)) for c in range(va.NUM_CHIPS)], [(c, prover.upload(m)) for c, m in fib25.preprocessed()])
    pinned = [va.pinned_copy(fib25.main_trace(c)) for c in range(va.NUM_CHIPS)]
    view = pinned[0][1:, :]
    first_row = pinned[0][1].copy()
    got = prover.prove([prover.upload(m) for m in pinned], [(c, prover.upload(m)) for c, m in fib25.preprocessed()])
    assert first_mismatch(got.words, ref.words) is None
    del pinned
    gc.collect()
    assert np.array_equal(view[0], first_row)  # the view keeps the block alive
    assert va.lib().vgpu_host_alloc(ctypes.c_uint64(0), ctypes.byref(ctypes.c_void_p())) == -1


@pytest.mark.gpu
def test_uploads_beside_a_running_proof(machine, rc):
    """A host prepares the next segment while the previous proof runs on the same context: vgpu_trace_upload then copies on a stream of
    its own (ordered behind the kernels that may still read recycled pool blocks); every proof equals the one proved from idle uploads."""
    p = va.Prover(machine, rc)
    ws = [va.Workload.fib(2000 + 7 * k) for k in range(4)]
    prep = [(c, p.upload(m)) for c, m in ws[0].preprocessed()]
    want = [p.prove([p.upload(w.main_trace(c)) for c in range(va.NUM_CHIPS)], prep).words.copy() for w in ws]
    host = [[va.pinned_copy(w.main_trace(c)) if k % 2 else w.main_trace(c) for c in range(va.NUM_CHIPS)] for k, w in enumerate(ws)]
    ticket, got = None, []
    for rep in range(3):
        for k in range(len(ws)):
            staged = [p.upload(m) for m in host[k]]  # while the previous ticket is still running
            if ticket is not None:
                got.append(ticket.wait().words.copy())
            ticket = p.prove_async(staged, prep)
    got.append(ticket.wait().words.copy())
    for i, g in enumerate(got):
        assert first_mismatch(g, want[i % len(ws)]) is None, i


@pytest.mark.gpu
def test_traces_of_another_context_of_the_same_device(machine, rc):
    """A side context uploads the operation logs and generates segment i+1's traces while the proving context is busy with segment i:
    vgpu_prove(_async) accepts those handles (same device), waits for the side context's queued work, and gives the same proofs."""
    p, side = va.Prover(machine, rc), va.Prover(machine, rc)
    ws = [va.Workload.fib(1500 + 11 * k) for k in range(3)]
    small = [i for i in range(va.NUM_CHIPS) if i not in va.GENERATED_CHIPS]
    prep = [(c, p.upload(m)) for c, m in ws[0].preprocessed()]
    want = [p.prove([p.upload(w.main_trace(c)) for c in range(va.NUM_CHIPS)], prep).words.copy() for w in ws]
    ticket, got = None, []
    for rep in range(2):
        for w in ws:
            log = side.upload_oplog(w.oplog())
            tr = {c: side.generate_trace(log, c) for c in va.GENERATED_CHIPS}
            tr.update({c: side.upload(w.main_trace(c)) for c in small})
            if ticket is not None:
                got.append(ticket.wait().words.copy())
            ticket = p.prove_async([tr[c] for c in range(va.NUM_CHIPS)], prep, keep=(log, side))
            del tr, log  # the ticket keeps what it works on alive
    got.append(ticket.wait().words.copy())
    for i, g in enumerate(got):
        assert first_mismatch(g, want[i % len(ws)]) is None, i
    sync = p.prove([side.upload(ws[0].main_trace(c)) for c in range(va.NUM_CHIPS)], prep)
    assert first_mismatch(sync.words, want[0]) is None


@pytest.mark.gpu
def test_proofs_on_a_busy_context_queue_up(machine, rc):
    """One proof at a time per prover context, and callers QUEUE (the reference's `Machine: Sync`, machine/src/machine.rs:13: several threads may call
    prove on one machine): three vgpu_prove_async on one context while tickets are outstanding are served one after the other, each the right proof."""
    p = va.Prover(machine, rc)
    w = va.Workload.fib(3000)
    main = [p.upload(w.main_trace(c)) for c in range(va.NUM_CHIPS)]
    prep = [(c, p.upload(m)) for c, m in w.preprocessed()]
    want = p.prove(main, prep).words.copy()
    tickets = [p.prove_async(main, prep) for _ in range(3)]
    for t in reversed(tickets):  # waited for in another order than issued
        assert first_mismatch(t.wait().words, want) is None


@pytest.mark.gpu
def test_cached_preprocessed_commitment_follows_the_traces(machine, rc):
    """The commitment to the preprocessed traces (program ROM, range table: basic/src/lib.rs:189-201) is kept across proofs while the SAME
    uploaded traces come back, and recomputed for any other set: a prover that alternates between two programs gives, every time, the proof a
    fresh prover (which recomputes it every time, the default) gives: hit after a miss, miss after a hit, a re-upload of equal contents."""
    p = va.Prover(machine, rc)
    p.set_prep_cache(True)  # off by default
    proofs = {}
    for n in (25, 40, 25, 40):
        w = va.Workload.fib(n)
        if n not in proofs:  # keep the uploads of a program for its second round: the cache keys on them
            proofs[n] = {"main": [p.upload(w.main_trace(c)) for c in range(va.NUM_CHIPS)], "prep": [(c, p.upload(m)) for c, m in w.preprocessed()], "w": w}
            fresh = va.Prover(machine, rc)
            proofs[n]["want"] = fresh.prove([fresh.upload(w.main_trace(c)) for c in range(va.NUM_CHIPS)], [(c, fresh.upload(m)) for c, m in w.preprocessed()]).words.copy()
        e = proofs[n]
        assert first_mismatch(p.prove(e["main"], e["prep"]).words, e["want"]) is None
        assert first_mismatch(p.prove(e["main"], e["prep"]).words, e["want"]) is None  # the hit
    e = proofs[25]
    again = [(c, p.upload(m)) for c, m in e["w"].preprocessed()]  # equal contents, new uploads: a miss, same proof
    assert first_mismatch(p.prove(e["main"], again).words, e["want"]) is None


def test_shader_clock_probe_reads_a_plausible_clock():
    """vgpu_shader_clock_probe (bench.py's measurement aid): one wave, shader cycles against the 100 MHz wall clock."""
    hz = [va.shader_clock_hz(0, 4096) for _ in range(3)]
    assert all(0.5e9 < h < 3.0e9 for h in hz), hz
    with pytest.raises(va.VgpuError):
        va.shader_clock_hz(0, 0)
