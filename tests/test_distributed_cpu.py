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
