"""§8(f)-4 completed: ONE proof over W ranks (vgpu_prove_sharded / vgpu_prove_sharded_local) must be, word for word, the proof
vgpu_prove produces on one GPU AND the oracle's proof of the same traces (computed here for the small workloads, the committed fixture
tests/golden/full_c2_fib149794.json at the headline size): a regression that moves the single-GPU and the sharded prover together is caught.  `world` prover contexts on this box's single GPU
stand in for the ranks (exchanges = device-to-device copies); over RCCL the same phases run with one rank per process — a world of one
exercises that fabric here.

Run on the MI355X box with `pytest -m gpu`.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import valida_amd as va
from conftest import first_mismatch
from oracle import pyoracle as po  # checker only

pytestmark = pytest.mark.gpu


def where(words, at, n_chips=14):
    """Name the region of the flat "VPF1" proof (DESIGN.md "Proof wire format") word `at` lies in: which stage of the sharded prover went wrong."""
    if at < 2:
        return "header"
    if at < 26:
        return ["main root", "permutation root", "quotient root"][(at - 2) // 8]
    pos = 26
    for chip in range(n_chips):
        start = pos
        pos += 1
        for vec in range(5):
            ln = int(words[pos])
            pos += 1 + 5 * ln
            if at < pos:
                return "chip %d: opened vector %d (0/1 main at zeta / zeta g, 2/3 permutation, 4 quotient chunks)" % (chip, vec)
        pos += 5
        if at < pos:
            return "chip %d: cumulative sum" % chip
        assert start < pos
    tail = at - pos
    n_commits = int(words[pos])
    if tail < 1 + 8 * n_commits:
        return "FRI commit-phase root of layer %d" % ((tail - 1) // 8)
    return "proof tail word %d (after the %d FRI roots: query openings / final polynomial / proof of work)" % (tail, n_commits)


def assert_same_proof(got, want):
    d = first_mismatch(got, want)
    if d is not None:
        at = int(np.nonzero(np.asarray(got).ravel()[: len(want)] != np.asarray(want).ravel()[: len(got)])[0][0]) if len(got) == len(want) else 0
        raise AssertionError("%s -- first difference in: %s" % (d, where(want, at)))


def assert_oracle_proof(words, mt, prep, rc, hash_kind=0, log_blowup=1):
    """The oracle proves the same traces now: every word of the sharded proof against it (not only against the single-GPU product proof)."""
    po.set_mmcs_hash(hash_kind, rc)
    try:
        ref = po.prove_basic(mt, prep[0][1], prep[1][1], rc, log_blowup=log_blowup)
    finally:
        po.set_mmcs_hash(0)
    assert_same_proof(words, ref.words)


def assert_fixture(proof, mt, name):
    """Committed fixture of the oracle's proof (tests/golden/make_golden.py --full): commitments, transcript, length, sha256 of the bytes."""
    with open(os.path.join(os.path.dirname(__file__), "golden", name)) as f:
        g = json.load(f)
    assert hashlib.sha256(b"".join(m.tobytes() for m in mt)).hexdigest() == g["traces_sha256"]
    assert [int(x) for x in proof.words[2:26]] == g["commitments"]
    assert int(proof.words.size) == g["proof_words"]
    assert hashlib.sha256(proof.bytes()).hexdigest() == g["proof_sha256"]


def single_and_inputs(machine, rc, workload, **cfg):
    p = va.Prover(machine, rc, **cfg)
    mt, prep = workload.main_traces(), workload.preprocessed()
    proof = p.prove([p.upload(m) for m in mt], [(c, p.upload(m)) for c, m in prep])
    return p, mt, prep, proof


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_sharded_proof_of_fib25_is_the_single_gpu_proof(machine, rc, fib25, world):
    """prove_fibonacci (basic/tests/test_prover.rs:473-486) with every chip above 4 * world LDE rows sharded: cpu / mem / add / mul / range /
    program in row ranges (the two preprocessed traces included: halo of all three LDEs), the height-1 chips replicated and injected at the
    subtree roots (world 2) or above them (world 4, 8); every FRI layer down to 4 * world elements sharded."""
    p0, mt, prep, single = single_and_inputs(machine, rc, fib25)
    provers = [p0] + [va.Prover(machine, rc) for _ in range(world - 1)]
    sharded = va.prove_sharded_local(provers, mt, prep, log_min_sharded=2)
    assert_same_proof(sharded.words, single.words)
    assert_oracle_proof(sharded.words, mt, prep, rc)


@pytest.mark.parametrize("world,log_min", [(4, 12), (2, 10)])
def test_sharded_proof_with_sharded_and_replicated_chips(machine, rc, world, log_min):
    """fib(582): cpu 2^12 / mem 2^14 / add 2^12 rows sharded (shards of >= 1024 rows: the matrix-core opened values, strided NTT passes),
    mul / range / program and the height-1 chips below the threshold: computed whole by every rank, row ranges of their LDEs hashed into the subtrees."""
    w = va.Workload.fib(582)
    p0, mt, prep, single = single_and_inputs(machine, rc, w)
    provers = [p0] + [va.Prover(machine, rc) for _ in range(world - 1)]
    sharded = va.prove_sharded_local(provers, mt, prep, log_min_sharded=log_min)
    assert_same_proof(sharded.words, single.words)
    assert_oracle_proof(sharded.words, mt, prep, rc)


def test_sharded_proof_of_the_alu_workload(machine, rc):
    """C4's program (add, sub, xor / and / or, lt: six busy chips, wide bitwise and lt traces) over four ranks."""
    w = va.Workload.alu(300)
    p0, mt, prep, single = single_and_inputs(machine, rc, w)
    provers = [p0] + [va.Prover(machine, rc) for _ in range(3)]
    sharded = va.prove_sharded_local(provers, mt, prep, log_min_sharded=6)
    assert_same_proof(sharded.words, single.words)
    assert_oracle_proof(sharded.words, mt, prep, rc)


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_proof_at_the_headline_size(machine, rc, world):
    """C2 (fib 149 794: cpu 2^20, mem 2^22 rows; the single-GPU proof of these traces is bit-matched to the oracle in test_gpu_parity) with the
    default threshold: cpu / mem / add sharded (row ranges of 2^18 .. 2^22 rows), everything else replicated, FRI layers sharded down to 2^12."""
    w = va.Workload.fib(149794)
    p0, mt, prep, single = single_and_inputs(machine, rc, w)
    provers = [p0] + [va.Prover(machine, rc) for _ in range(world - 1)]
    sharded = va.prove_sharded_local(provers, mt, prep)
    assert_same_proof(sharded.words, single.words)
    assert_fixture(sharded, mt, "full_c2_fib149794.json")  # the ORACLE's proof of these traces, pinned


def test_sharded_proof_with_the_poseidon_mmcs(machine, rc, fib25):
    p0, mt, prep, single = single_and_inputs(machine, rc, fib25, hash_kind=va.HASH_POSEIDON16)
    provers = [p0, va.Prover(machine, rc, hash_kind=va.HASH_POSEIDON16)]
    sharded = va.prove_sharded_local(provers, mt, prep, log_min_sharded=3)
    assert_same_proof(sharded.words, single.words)
    assert_oracle_proof(sharded.words, mt, prep, rc, hash_kind=va.HASH_POSEIDON16)


def test_sharded_proof_over_the_rccl_fabric_world_of_one(machine, rc, fib25):
    """The RCCL realisation (one rank per process) on this box's single GPU: a world of one runs every phase through the communicator's
    all-gather / grouped send-recv call paths."""
    p, mt, prep, single = single_and_inputs(machine, rc, fib25)
    comm = va.Comm(p, va.Comm.unique_id(), 0, 1)
    sharded = comm.prove_sharded([p.upload(m) for m in mt], [(c, p.upload(m)) for c, m in prep], log_min_sharded=4)
    assert_same_proof(sharded.words, single.words)
    assert_oracle_proof(sharded.words, mt, prep, rc)


@pytest.mark.parametrize("world,log_blowup,log_min", [(1, 2, 2), (2, 2, 2), (4, 2, 3), (8, 2, 2), (4, 3, 2), (2, 3, 2), (8, 3, 3)])
def test_sharded_proof_with_a_larger_blowup(machine, rc, fib25, world, log_blowup, log_min):
    """C3's configuration (4x blowup) and 8x: the quotient domain is the first L >> (log_blowup - 1) storage rows of every LDE (machine/src/quotient.rs:41-47
    takes it as a strided view), i.e. the row ranges of the first world >> (log_blowup - 1) ranks: they evaluate the quotient (halo among
    themselves once there are more than two), every rank receives its columns of the chunks.  Also world < 2^(log_blowup - 1): one rank, part of its range."""
    p0, mt, prep, single = single_and_inputs(machine, rc, fib25, log_blowup=log_blowup)
    provers = [p0] + [va.Prover(machine, rc, log_blowup=log_blowup) for _ in range(world - 1)]
    sharded = va.prove_sharded_local(provers, mt, prep, log_min_sharded=log_min)
    assert_same_proof(sharded.words, single.words)
    assert_oracle_proof(sharded.words, mt, prep, rc, log_blowup=log_blowup)


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_proof_of_c3s_configuration_at_2_16(machine, rc, world):
    """fib(9359): cpu 2^16 / mem 2^18 rows with the 4x blowup of C3 (LDEs of 2^18 .. 2^20 rows sharded, strided NTT passes, matrix-core opened values)."""
    w = va.Workload.fib(9359)
    p0, mt, prep, single = single_and_inputs(machine, rc, w, log_blowup=2)
    provers = [p0] + [va.Prover(machine, rc, log_blowup=2) for _ in range(world - 1)]
    sharded = va.prove_sharded_local(provers, mt, prep)
    assert_same_proof(sharded.words, single.words)
    assert_oracle_proof(sharded.words, mt, prep, rc, log_blowup=2)


def test_sharded_proof_of_c3_at_full_size(machine, rc):
    """C3 itself (fib 599 183: cpu 2^22, mem 2^24 rows, 4x blowup — the proof whose 116 GB of algorithmic traffic is the reason to shard) over 8
    ranks: the oracle's proof of these traces, by the committed fixture."""
    w = va.Workload.fib(599183)
    mt, prep = w.main_traces(), w.preprocessed()
    provers = [va.Prover(machine, rc, log_blowup=2) for _ in range(8)]
    sharded = va.prove_sharded_local(provers, mt, prep)
    assert_fixture(sharded, mt, "full_c3_fib599183_blowup4.json")


# ---- the traces themselves sharded (round-3 verdict, item 6): every sharded chip hands in only its rows [rank n / W, (rank + 1) n / W) -----------------
@pytest.mark.parametrize("world", [2, 4, 8])
def test_row_range_inputs_fib25_every_chip_split(machine, rc, fib25, world):
    """prove_fibonacci with every chip above 4 * world LDE rows handed in as row ranges (the chips with preprocessed traces included: their
    permutation traces read a row range of the whole preprocessed trace): local scans + ONE exchange of the ranks' totals reproduce
    generate_permutation_trace's running sums and cumulative sums (machine/src/chip.rs:176-205), the commitment rounds deal rows into columns."""
    p0, mt, prep, single = single_and_inputs(machine, rc, fib25)
    provers = [p0] + [va.Prover(machine, rc) for _ in range(world - 1)]
    split = [va.sharded_trace_is_split(world, m.shape[0], 1, 2) for m in mt]
    assert any(split) and not all(split)
    sharded = va.prove_sharded_rows_local(provers, mt, prep, log_min_sharded=2)
    assert_same_proof(sharded.words, single.words)
    assert_oracle_proof(sharded.words, mt, prep, rc)


@pytest.mark.parametrize("world,log_min,log_blowup", [(4, 12, 1), (2, 10, 1), (8, 8, 2)])
def test_row_range_inputs_with_split_and_whole_chips(machine, rc, world, log_min, log_blowup):
    """fib(582): cpu / mem / add as row ranges, the small chips whole — and with C3's 4x blowup over eight ranks."""
    w = va.Workload.fib(582)
    p0, mt, prep, single = single_and_inputs(machine, rc, w, log_blowup=log_blowup)
    provers = [p0] + [va.Prover(machine, rc, log_blowup=log_blowup) for _ in range(world - 1)]
    sharded = va.prove_sharded_rows_local(provers, mt, prep, log_min_sharded=log_min)
    assert_same_proof(sharded.words, single.words)
    assert_oracle_proof(sharded.words, mt, prep, rc, log_blowup=log_blowup)


def test_row_range_inputs_of_the_alu_workload_and_the_poseidon_mmcs(machine, rc):
    w = va.Workload.alu(300)
    p0, mt, prep, single = single_and_inputs(machine, rc, w, hash_kind=va.HASH_POSEIDON16)
    provers = [p0] + [va.Prover(machine, rc, hash_kind=va.HASH_POSEIDON16) for _ in range(3)]
    sharded = va.prove_sharded_rows_local(provers, mt, prep, log_min_sharded=6)
    assert_same_proof(sharded.words, single.words)
    assert_oracle_proof(sharded.words, mt, prep, rc, hash_kind=va.HASH_POSEIDON16)


@pytest.mark.parametrize("world", [2, 8])
def test_row_range_inputs_at_the_headline_size(machine, rc, world):
    """C2 with cpu / mem / add handed in as row ranges: per-rank trace memory 1 / world; the oracle's proof by the committed fixture."""
    w = va.Workload.fib(149794)
    mt, prep = w.main_traces(), w.preprocessed()
    provers = [va.Prover(machine, rc) for _ in range(world)]
    sharded = va.prove_sharded_rows_local(provers, mt, prep)
    assert_fixture(sharded, mt, "full_c2_fib149794.json")
    # what a rank was handed: its share of the three big chips, the small ones whole
    mats, full = va.row_ranges(mt, 1, world)
    assert sum(m.nbytes for m in mats) < 1.02 * sum(m.nbytes for m in mt) / world + sum(m.nbytes for m, f in zip(mt, full) if not va.sharded_trace_is_split(world, f))


def test_row_range_inputs_are_validated(machine, rc, fib25):
    mt, prep = fib25.main_traces(), fib25.preprocessed()
    provers = [va.Prover(machine, rc) for _ in range(2)]
    real = va.row_ranges

    def wrong(main_traces, rank, world, log_blowup=1, log_min_sharded=12):
        mats, full = real(main_traces, rank, world, log_blowup, log_min_sharded)
        mats[0] = main_traces[0]  # the whole cpu trace where a row range is due
        return mats, full

    va.row_ranges = wrong
    try:
        with pytest.raises(va.VgpuError, match="must hand in its row range"):
            va.prove_sharded_rows_local(provers, mt, prep, log_min_sharded=2)
    finally:
        va.row_ranges = real


def test_sharded_proof_refuses_what_it_does_not_implement(machine, rc, fib25):
    mt, prep = fib25.main_traces(), fib25.preprocessed()
    provers = [va.Prover(machine, rc) for _ in range(3)]
    with pytest.raises(va.VgpuError, match="power of two"):
        va.prove_sharded_local(provers, mt, prep)
    p = va.Prover(machine, rc)
    with pytest.raises(va.VgpuError, match="distinct"):
        va.prove_sharded_local([p, p], mt, prep)


def test_device_proofs_pass_the_products_own_verifier(machine, rc, fib25):
    """prove on the device, verify with the library's host-side Machine::verify (vgpu_verify; preprocessed commitment recomputed on the host by
    vgpu_host_commit_root and equal to the one the device committed): the pair a host of this library uses, no oracle involved."""
    for w in (fib25, va.Workload.alu(40)):
        p0, mt, prep, single = single_and_inputs(machine, rc, w)
        pc = va.host_commit_root([m for _, m in prep], rc)
        assert first_mismatch(pc, single.transcript[0:8]) is None
        assert va.verify(machine, rc, single.words, pc) is None
        bad = single.words.copy()
        bad[40] = (int(bad[40]) + 1) % va.P
        assert va.verify(machine, rc, bad, pc) is not None


def test_local_fabric_staged_copies_give_the_same_proof():
    """LocalFabric's fallback for device pairs WITHOUT a peer path (exchanges staged through page-locked host memory, csrc/host/fabric.hpp: copy_staged) cannot
    occur on a 1-GPU box; the test-only hook VGPU_TESTING=1 VGPU_FAILPOINT=local_stage_copies@0 sends every exchange of a sharded proof through it.  In a
    process of its own (the hook is read once): the sharded proof of fib(582) over four contexts must be the oracle's."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import hashlib, json, sys; sys.path.insert(0, %r); import valida_amd as va\n"
            "g = json.load(open(%r)); w = va.Workload.fib(g['n']); mt, prep = w.main_traces(), w.preprocessed()\n"
            "m, rc = va.Machine.basic(), va.poseidon_round_constants()\n"
            "ps = [va.Prover(m, rc) for _ in range(4)]\n"
            "pr = va.prove_sharded_local(ps, mt, prep, log_min_sharded=10)\n"
            "print(json.dumps({'sha': hashlib.sha256(pr.bytes()).hexdigest()}))\n") % (root, os.path.join(root, "tests", "golden", "fib582_oracle.json"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VGPU_TESTING="1", VGPU_FAILPOINT="local_stage_copies@0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "test hook local_stage_copies" in r.stderr  # the staged path really ran
    with open(os.path.join(root, "tests", "golden", "fib582_oracle.json")) as f:
        g = json.load(f)
    assert json.loads(r.stdout.strip().splitlines()[-1])["sha"] == g["proof_sha256"]
