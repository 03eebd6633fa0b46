"""Reproduction harness for the round-3 driver-box hang: the world-4 one-rank-per-process proof of fib(582) in a
loop, every iteration through tests/test_zz_sharded_multiprocess_gpu.py's own _run (per-rank reports, exit codes, stderr tails).
    python tools/mp_loop.py [iterations] [world] > gpurun_out/r04_mp_loop.log
Run it under `taskset -c 0-15` to mimic the driver box's 16 cores."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_zz_sharded_multiprocess_gpu as t

    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    bad = 0
    print("cores available to this process: %d" % len(os.sched_getaffinity(0)), flush=True)
    for i in range(iters):
        t0 = time.time()
        res = t._run(world, "fib582_oracle.json", 12 if world == 4 else 10)
        ok = all("sha" in res[r] and [res[r]["commitments"], res[r]["words"], res[r]["sha"]] == res[r]["want"] for r in range(world))
        print("iteration %2d: %s in %5.1f s; prove_s per rank %s; exit codes %s" % (i, "ok" if ok else "FAILED", time.time() - t0, [res[r].get("prove_s") for r in range(world)],
                                                                               [res[r]["exitcode"] for r in range(world)]), flush=True)
        if not ok:
            bad += 1
            print(t._describe(res), flush=True)
    print("%d of %d iterations failed" % (bad, iters), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
