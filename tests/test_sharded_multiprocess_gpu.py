"""ONE proof over W ranks with one rank per PROCESS: the C++ sharded prover (valida_amd/csrc/host/sharded_prover.cpp)
driven through the caller-supplied fabric of the C ABI (vgpu_fabric_t / vgpu_prove_sharded_fabric) by 2 and 4 processes that share this
box's one MI355X, their exchanges carried by torch.distributed gloo.  Every rank's proof must be the ORACLE's proof of the traces (the
committed fixtures tests/golden/fib582_oracle.json and full_c2_fib149794.json).  What this does not measure is xGMI: the transport is host
memory; on a multi-GPU node the same prover runs over the RCCL fabric (vgpu_prove_sharded), which only the driver's scaling run can reach.

Run on the MI355X box with `pytest -m gpu`.
"""
import hashlib
import json
import os
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, fixture, log_min, sabotage):
    sys.path.insert(0, ROOT)
    try:
        import torch.distributed as dist

        import valida_amd as va

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        with open(os.path.join(ROOT, "tests", "golden", fixture)) as f:
            g = json.load(f)
        w = va.Workload.fib(g["n"])
        mt, prep = w.main_traces(), w.preprocessed()
        assert hashlib.sha256(b"".join(m.tobytes() for m in mt)).hexdigest() == g["traces_sha256"]
        p = va.Prover(va.Machine.basic(), va.poseidon_round_constants(), device=0)  # every process its own context on the shared GPU
        if sabotage is not None and rank == sabotage[0]:
            mt = list(mt)
            if sabotage[1] == "width":   # this rank alone fails its own validation (before its first exchange)
                mt[3] = mt[3][:, :-1].copy()
            else:                        # a valid trace, but of another height than its peers': only the ranks together can notice
                mt[3] = mt[3][: mt[3].shape[0] // 2].copy()
        fab = va.Fabric.over_torch_distributed(dist)
        dmain, dprep = [p.upload(m) for m in mt], [(c, p.upload(m)) for c, m in prep]
        try:
            proof = fab.prove_sharded(p, dmain, dprep, log_min_sharded=log_min)
            res = {"commitments": [int(x) for x in proof.words[2:26]], "words": int(proof.words.size), "sha": hashlib.sha256(proof.bytes()).hexdigest(),
                   "want": [g["commitments"], g["proof_words"], g["proof_sha256"]]}
        except va.VgpuError as e:
            res = {"error": str(e)}
        res["callback_errors"] = [repr(e) for e in fab.errors]
        q.put((rank, res))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 - reported to the parent, which fails the test
        q.put((rank, {"crash": "%s: %s" % (type(e).__name__, e)}))


def _run(world, fixture, log_min, sabotage=None, timeout=600):
    port = 33000 + (os.getpid() % 2000) + 11 * world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, fixture, log_min, sabotage)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = dict(q.get(timeout=timeout) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    return results


@pytest.mark.parametrize("world,log_min", [(2, 10), (4, 12)])
def test_one_rank_per_process_fib582_is_the_oracles_proof(world, log_min):
    res = _run(world, "fib582_oracle.json", log_min)
    for rank in range(world):
        r = res[rank]
        assert "crash" not in r and "error" not in r and not r["callback_errors"], (rank, r)
        assert [r["commitments"], r["words"], r["sha"]] == r["want"], "rank %d" % rank


@pytest.mark.parametrize("world", [2, 4])
def test_one_rank_per_process_at_the_headline_size(world):
    """C2 (fib 149 794: cpu 2^20, mem 2^22 rows), cpu / mem / add sharded, FRI layers sharded down to 2^12: every process's proof has the
    sha256 of the oracle's proof of these traces."""
    res = _run(world, "full_c2_fib149794.json", 12, timeout=900)
    for rank in range(world):
        r = res[rank]
        assert "crash" not in r and "error" not in r and not r["callback_errors"], (rank, r)
        assert [r["commitments"], r["words"], r["sha"]] == r["want"], "rank %d" % rank


def test_a_rank_that_fails_alone_fails_every_rank():
    """The failure protocol inside the real prover: rank 1 of 2 passes an add trace of the wrong width and throws in its
    own validation; rank 0 must come back with an error naming rank 1, not block in an exchange rank 1 never enters."""
    res = _run(2, "fib582_oracle.json", 10, sabotage=(1, "width"), timeout=300)
    assert "error" in res[1] and "width mismatch" in res[1]["error"], res[1]
    assert "error" in res[0] and "rank 1 failed" in res[0]["error"], res[0]


def test_ranks_called_with_different_shapes_all_refuse():
    """Each rank's traces are valid on their own, but rank 1's add trace has another height: the exchanges would be sized differently on the two
    sides.  The shapes are compared across the ranks before the first sized exchange and every rank refuses."""
    res = _run(2, "fib582_oracle.json", 10, sabotage=(1, "height"), timeout=300)
    for rank in (0, 1):
        assert "error" in res[rank] and "other shapes" in res[rank]["error"], res[rank]
