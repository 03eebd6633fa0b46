"""world_size-2 test of the N>1 path on CPU (gloo): segments are independent proofs, the only exchange is
the all-gather of each segment's three commitment roots (bench.exchange_roots).  On the MI355X node the
same function runs over RCCL/xGMI with the nccl backend."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import bench
    import valida_amd as va
    from oracle import pyoracle as po

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    # each rank proves ITS segment (here with the CPU oracle standing in for the device prover)
    n = bench.segment_loop_bound(12, rank)
    w = va.Workload.fib(n)
    assert w.cpu_height == 1 << 12
    prep = w.preprocessed()
    res = po.prove_basic(w.main_traces(), prep[0][1], prep[1][1], va.poseidon_round_constants(), num_queries=4)
    roots = bench.exchange_roots(dist, torch, res.words[2:26], torch.device("cpu"))
    # a batch of K proofs per rank travels in the same single collective (bench.run_steps): 24 words per proof
    batch = bench.exchange_roots(dist, torch, np.concatenate([res.words[2:26], res.words[2:26][::-1]]), torch.device("cpu"))
    assert tuple(batch.shape) == (world, 48) and batch[rank, :24].tolist() == [int(x) for x in res.words[2:26]]
    assert batch[:, 24:].tolist() == [r[::-1] for r in roots.numpy().tolist()]
    q.put((rank, n, [int(x) for x in res.words[2:26]], roots.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_segments_exchange_their_roots():
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, own0, all0), (r1, n1, own1, all1) = results
    assert n0 != n1 and own0 != own1  # distinct segments, distinct commitments
    assert all0 == all1 == [own0, own1]  # every rank holds every segment's caps, in rank order


# ---- ONE commitment round of one proof sharded over the ranks (the protocol of
# valida_amd/csrc/host/sharded.hpp with the CPU oracle standing in for the device kernels and gloo for RCCL) ----------------
def _sharded_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from oracle import pyoracle as po

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(11)  # every rank sees the same matrices (it extends only its own columns)
    P = 2013265921
    mats = [rng.integers(0, P, (64, 5), dtype=np.uint32), rng.integers(0, P, (256, 3), dtype=np.uint32), rng.integers(0, P, (64, 2), dtype=np.uint32),
            rng.integers(0, P, (8, 7), dtype=np.uint32), rng.integers(0, P, (1, 3), dtype=np.uint32)]
    # 1. column shards: global column g belongs to rank g mod W; LDE (blowup 2, committed row order) of the own columns only
    base, own, lde = 0, [], []
    for m in mats:
        cols = [c for c in range(m.shape[1]) if (base + c) % world == rank]
        own.append(cols)
        lde.append(po.committed_lde(m[:, cols], 1, 31) if cols else np.zeros((2 * m.shape[0], 0), dtype=np.uint32))
        base += m.shape[1]
    # 2. all-to-all: rank s gets rows [s L / W, (s + 1) L / W) of every column
    send = [[lde[i][s * lde[i].shape[0] // world:(s + 1) * lde[i].shape[0] // world] for i in range(len(mats))] for s in range(world)]
    gathered = [None] * world
    dist.all_gather_object(gathered, send)      # gloo has no all-to-all: every rank picks its row of the matrix of messages
    recv = [gathered[src][rank] for src in range(world)]
    # 3. reassemble the row range of every matrix in commit column order and build the subtree over it
    shard, base = [], 0
    for i, m in enumerate(mats):
        rows = 2 * m.shape[0] // world
        full = np.zeros((rows, m.shape[1]), dtype=np.uint32)
        nxt = [0] * world
        for c in range(m.shape[1]):
            src = (base + c) % world
            full[:, c] = recv[src][i][:, nxt[src]]
            nxt[src] += 1
        shard.append(full)
        base += m.shape[1]
    sub = po.mmcs_root(shard)
    # 4. all-gather of the subtree roots; the top level(s): C(root_0, root_1) for two ranks (no matrix is shorter than W here)
    roots = [None] * world
    dist.all_gather_object(roots, sub)
    assert world == 2
    root = po.compress(roots[0], roots[1])
    q.put((rank, [int(x) for x in root], [int(x) for x in po.commit_root(mats)]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_commit_protocol_over_gloo():
    world, port = 2, 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, root, want in results:
        assert root == want, rank  # every rank ends with the root pcs.commit_batches gives on one machine
