#!/usr/bin/env python
"""Headline benchmark: proofs/sec of Machine::prove on the 2^20-row Fibonacci trace (BASELINE.json
configs[1] = workload C2 of SURVEY.md §8), one proof per GPU per step.

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete proof per rank (from HBM-resident main traces to the assembled proof words on
the host) + one RCCL all-gather of that proof's three 32-byte commitment roots over xGMI when N > 1, issued
as soon as the rank's proof is complete (segments are independent).  Rank 0 prints ONE JSON
line.  Scaling is weak (one segment per GPU).

Beside the headline the line carries (N = 1) the latency of a lone proof, the PCIe-inclusive and operation-log legs, the CPU baseline
`one_proof_over_w_ranks_on_this_gpu` (ONE proof sharded over W prover contexts of this device) and
`proof_checked_by_vgpu_verify`: the last timed proof through the library's own host-side Machine::verify.  With N > 1
the ranks finish by proving one segment TOGETHER through vgpu_prove_sharded (RCCL) — after the JSON line, reported on stderr.
"""
import argparse
import json
import os
import hashlib
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Fibonacci loop bound giving 17 + 7n = 2^20 - 1 executed cycles.
FIB_N = {22: 599183, 20: 149794, 19: 74895, 18: 37446, 17: 18722, 16: 9359, 14: 2338, 12: 582}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_proof(shapes, log_blowup=1):
    """SURVEY.md §8(d) B_alg: compulsory HBM traffic of one proof, each array once per logical pass."""
    b, q, D = 1 << log_blowup, 2, 5
    main = sum(n * w for n, w, _ in shapes)
    perm = sum(n * 5 * (m + 1) for n, _, m in shapes)
    quot = sum(n * 10 for n, _, _ in shapes)
    prep = 32 * 7 + 256
    E = main + perm + quot
    sum_n = sum(n for n, _, _ in shapes)
    L = b * max(n for n, _, _ in shapes)
    heights = sorted({n for n, _, _ in shapes})
    total = 4 * E + 4 * b * E + 4 * b * E + 3 * (4 * 32 * L) + 4 * (main + perm) + 4 * q * (main + perm + prep) + 4 * D * q * sum_n
    total += (4 * D * q * sum_n + 4 * quot) + 4 * b * E + 4 * D * sum(b * n for n in heights) + 2 * (4 * D * 2 * L) + 2 * (2 * (L // 2) * 32 * 2)
    return float(total)


# Integer-VALU facts of the MI355X, measured by tools/microbench.py (profiles/rNN_microbench.txt; these defaults are the r02
# numbers and are replaced by whatever the newest committed report says): G wave64-instr/s over the chip for the instruction
# classes the prover is made of, and the in-register ceiling of the product's own Keccak-f[1600].
MICROBENCH_DEFAULTS = {"full_rate": 1.03e12,   # v_add_u32 / v_sub_u32 / v_xor_b32 / v_bitop3_b32 / v_fma_f32: ~2.4 cycles per SIMD
                       "half_rate": 0.586e12,  # v_alignbit_b32, v_mul_lo/hi_u32, v_mad_u64_u32, v_min_u32, shifts: ~4.2 cycles per SIMD
                       "keccak_perm_per_s": 9.9e9, "clock_hz": 2.35e9, "source": "defaults (r02 measurement)"}
KECCAK_VALU_PER_PERM = 23 * 178 + 58  # kernels/merkle.hip: instructions of one digest-only permutation
KECCAK_FULL_RATE_PER_ROUND, KECCAK_HALF_RATE_PER_ROUND = 122, 56  # bitop3/xor vs alignbit per round
# The ISSUE BOUND every VALU roofline below is priced against (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on its
# SIMD-32, the half-rate class — v_alignbit, 32-bit multiplies, v_mad_u64_u32 — over 4; 256 CUs x 4 SIMDs at the 2.4 GHz maximum clock).
# No schedule of a given instruction mix can run faster than this; the measured isolated rates (tools/microbench.hip: 2.3-2.5 and
# 4.1-4.2 cycles) and the code's own in-register ceiling are reported beside it, never as `peak`.
GUIDE_SIMDS, GUIDE_CLOCK_HZ, GUIDE_FULL_CYCLES, GUIDE_HALF_CYCLES = 1024, 2.4e9, 2.0, 4.0


def issue_bound_per_s(full_per_unit, half_per_unit):
    """Units (permutations ...) per second when every SIMD does nothing but issue the unit's VALU instructions at the guide's rates; one
    wave-instruction serves 64 units (one per lane)."""
    return GUIDE_SIMDS * GUIDE_CLOCK_HZ * 64.0 / (full_per_unit * GUIDE_FULL_CYCLES + half_per_unit * GUIDE_HALF_CYCLES)

# kernels/poseidon_mmcs.hip (POSEIDON_HALF_PER_PERM / POSEIDON_FULL_PER_PERM): instructions of one Poseidon-16 permutation as the kernels run it
# (8 full rounds with the MDS layer as CRT blocks, 21 sparse partial rounds, one dense partial round), by issue class
POSEIDON_HALF_PER_PERM, POSEIDON_FULL_PER_PERM = 3945, 5264


def microbench_facts():
    """Peaks from the newest profiles/r*_microbench.txt (tools/microbench.py output), else the r02 defaults."""
    import glob
    import re

    facts = dict(MICROBENCH_DEFAULTS)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_microbench.txt")))
    if not files:
        return facts
    try:
        txt = open(files[-1]).read()
        rate = {m.group(1): float(m.group(2)) * 1e9 for m in re.finditer(r"^(v_\w+)\s.*peak\s+([0-9.]+) G wave-instr/s", txt, re.M)}
        full = [rate[k] for k in ("v_add_u32", "v_xor_b32", "v_bitop3_b32") if k in rate]
        half = [rate[k] for k in ("v_alignbit_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32") if k in rate]
        if full and half:
            facts["full_rate"], facts["half_rate"] = sum(full) / len(full), sum(half) / len(half)
        k = re.search(r"^keccak_f1600 digest(.*)$", txt, re.M)
        if k:
            perms = [float(x) * 1e9 for x in re.findall(r"([0-9.]+) G perm/s", k.group(1))]
            clocks = [float(x) * 1e9 for x in re.findall(r"([0-9.]+) GHz", k.group(1))]
            facts["keccak_perm_per_s"], facts["clock_hz"] = max(perms), clocks[perms.index(max(perms))]
        k = re.search(r"^poseidon16 chain(.*)$", txt, re.M)
        if k:
            facts["poseidon_perm_per_s"] = max(float(x) * 1e9 for x in re.findall(r"([0-9.]+) G perm/s", k.group(1)))
        facts["source"] = os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError, KeyError):
        pass
    return facts


def pmc_file():
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None, {}
    try:
        with open(files[-1]) as f:
            return os.path.relpath(files[-1], ROOT), json.load(f)["kernels"]
    except (KeyError, ValueError, OSError):
        return None, {}


def pmc_traffic_per_launch(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes (profiles/rNN_pmc.json, written
    by tools/summarize_prof.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command), or None.
    NOT measured in this run: the line says where it comes from (roofline.traffic_source)."""
    src, kernels = pmc_file()
    k = kernels.get(kernel)
    try:
        return (k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"], src) if k else (None, src)
    except KeyError:
        return None, src


def segment_loop_bound(log_rows, rank):
    """Independent segments: rank r proves fib with a distinct loop bound and the same padded shape."""
    n = FIB_N[log_rows] - rank
    assert 17 + 7 * n > (1 << (log_rows - 1)), "too many ranks for this trace size"
    return n


def exchange_roots(dist, torch, commitments, device, group=None):
    """The one collective of the path: all-gather of each segment's three 8-word Merkle roots (96 B per proof
    per rank; RCCL over xGMI with the nccl backend, gloo in the CPU tests).  `commitments` holds 24 words per
    proof this rank produced; returns a [world, len(commitments)] int64 array."""
    world = dist.get_world_size()
    local = torch.from_numpy(np.ascontiguousarray(commitments, dtype=np.int64).reshape(-1)).to(device)
    out = torch.zeros(local.numel() * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out.reshape(world, local.numel())


def roofline_object(name, stat, achieved_gbs, traffic, traffic_src, valu, steps):
    """The bench line's `roofline`: the BINDING roofline of the dominant kernel first (`bound` says which), the HBM figures the
    contract asks for always present under `hbm` (and they are the primary ones when the kernel is HBM-bound)."""
    launches, ms, nbytes, _ = stat
    hbm = {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
           "traffic_source": "%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)" % traffic_src,
           "algorithmic_bytes_per_launch": nbytes / launches if launches else None}
    common = {"kernel": name, "launches_per_step": launches / steps, "avg_launch_ms": ms / launches if launches else None}
    if valu is None:
        return dict(hbm, **common, limited_by="hbm", hbm=hbm)
    return dict(common, bound="valu", achieved=valu["achieved"], peak=valu["peak"], unit=valu["unit"], frac=valu["frac"], traffic=traffic,
                limited_by="integer VALU issue (31-bit modular / Keccak lane arithmetic; no MFMA form exists), not HBM: `hbm` holds the contract's HBM figures",
                peak_is=valu.get("peak_is"), hbm=hbm)


def sustained_object(sus, world, kernel, valu_roofline, b_alg):
    """The bench line's `sustained`: proofs/s and ms per step over the long region, its mean shader clock, and the dominant kernel's live
    roofline of that region — at the guide's 2.4 GHz (a bound whatever the clock does) and at the region's own measured clock (what the
    kernel makes of the cycles it was given)."""
    ms_step = sus["seconds"] / sus["steps"] * 1e3
    clk = sum(sus["clock"]) / len(sus["clock"]) if sus["clock"] else None
    out = {"proofs_per_s": world * sus["steps"] / sus["seconds"], "ms_per_step": ms_step, "seconds": sus["seconds"], "steps": sus["steps"],
           "shader_clock_GHz": None if clk is None else clk / 1e9,
           "shader_clock_GHz_min_max": None if clk is None else [min(sus["clock"]) / 1e9, max(sus["clock"]) / 1e9],
           "proof_hbm_frac": b_alg / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
    st = sus["prof"].get(kernel)
    if st and st[1] > 0:
        out["roofline"] = {"kernel": kernel, "avg_launch_ms": st[1] / st[0], "hbm_achieved_GBs": st[2] / (st[1] * 1e-3) / 1e9,
                           "hbm_frac": st[2] / (st[1] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        v = valu_roofline(st) if st[3] > 0 else None
        if v:
            out["roofline"].update({"bound": "valu", "achieved": v["achieved"], "unit": v["unit"], "peak_at_guide_clock": v["peak"], "frac_at_guide_clock": v["frac"],
                                    "frac_at_measured_clock": None if clk is None else v["frac"] * GUIDE_CLOCK_HZ / clk})
    return out


def cpu_baseline(log_rows, rc, headline_log_rows, mmcs_poseidon=False, scalar_too=False):
    """The oracle (CPU restatement of the reference's algorithm, C++/OpenMP on every host core) timed on the SAME workload the GPU
    is timed on — the headline 2^20-row segment itself by default, one proof, no extrapolation — in its FAST mode (oracle/fast.hpp: AVX2
    8-wide Montgomery BabyBear transforms with precomputed twiddles, a four-way AVX2 Keccak, batch inversions, constraint folding with
    precomputed powers of alpha: the techniques SURVEY.md L0 attributes to Plonky3's x86 backend; proof words identical to the scalar
    oracle's, tests/test_oracle_cpu.py).  Still a port ("kind": "port-simd"), not Plonky3: the constraint evaluation is scalar and its
    matrices are row-major std::vectors; the real prover may well be another small factor faster."""
    import subprocess

    # a process of its own: the measurement must not share heap, OpenMP pool or cores with the HIP runtime and the prover threads of THIS process
    # (measured: 13.3 s in-process against 4.1 s alone on the same 16 cores)
    env = dict(os.environ, ORACLE_MMCS="1" if mmcs_poseidon else "0")
    env.pop("VGPU_SPIN_WAIT", None)
    r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", str(FIB_N[log_rows])], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline process failed: " + r.stderr[-400:])
    got = json.loads(r.stdout.strip().splitlines()[-1])
    seconds, cores = got["seconds"], got["cores"]
    out = {
        "value": 1.0 / seconds,
        "unit": "proofs/s",
        "seconds_per_proof": seconds,
        "cores": cores,
        "kind": "port-simd",
        "sample": "oracle in fast mode (C++/OpenMP + AVX2 restatement, not Plonky3; a process of its own) proving ONE fib segment with 2^%d cpu rows (mem 2^%d) in %.1f s on %d host cores"
                  % (log_rows, log_rows + 2, seconds, cores),
        "proof_sha256": got["sha256"],
    }
    if scalar_too:  # --scalar-baseline: the same oracle in its scalar `% p` mode (the checker the parity tests use), re-timed in THIS run (about a minute)
        r2 = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", str(FIB_N[log_rows]), "scalar"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=3600)
        if r2.returncode == 0:
            g2 = json.loads(r2.stdout.strip().splitlines()[-1])
            out["scalar_port"] = {"seconds_per_proof": g2["seconds"], "cores": g2["cores"], "proof_sha256": g2["sha256"], "same_proof_as_fast_mode": g2["sha256"] == got["sha256"]}
        else:
            out["scalar_port"] = {"error": r2.stderr[-300:]}
    if log_rows != headline_log_rows:  # a smaller sample was asked for: say so, and give the linear-in-rows estimate separately
        out["note"] = "sample is 2^%d rows, not the 2^%d-row workload of `value`; linear scaling would give %.4f proofs/s" % (
            log_rows, headline_log_rows, 1.0 / (seconds * (1 << (headline_log_rows - log_rows))))
    return out


def sharded_leg_process(args):
    """`bench.py --sharded-leg` (started by every rank of an N > 1 run AFTER its contract line): rank 0's segment proved ONCE over all the
    ranks (vgpu_prove_sharded over RCCL / xGMI), the result on stderr.  Its own rendezvous (gloo on VGPU_SHARDED_LEG_PORT, for the
    communicator id only), its own prover context and communicator with a deadline on every collective."""
    import datetime
    import hashlib
    import torch
    import torch.distributed as dist
    import valida_amd as va

    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ["WORLD_SIZE"])
    res = {}
    try:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="gloo", init_method="tcp://%s:%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ["VGPU_SHARDED_LEG_PORT"]),
                                rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        if args.workload == "c3":
            args.log_rows = 22
        rc = va.poseidon_round_constants()
        prover = va.Prover(va.Machine.basic(), rc, log_blowup=2 if args.workload == "c3" else 1, device=local_rank,
                           hash_kind=va.HASH_POSEIDON16 if args.mmcs == "poseidon" else va.HASH_KECCAK256)
        ids = [va.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        cm = va.Comm(prover, ids[0], rank, world)
        cm.set_timeout_ms(int(os.environ.get("VGPU_BENCH_COLLECTIVE_TIMEOUT_MS", "60000")))
        w0 = va.Workload.alu(((1 << args.log_rows) - 8) // 9) if args.workload == "c4" else va.Workload.fib(segment_loop_bound(args.log_rows, 0))
        m0, p0 = w0.main_traces(), w0.preprocessed()
        dm, dp = [prover.upload(m) for m in m0], [(c, prover.upload(m)) for c, m in p0]
        pr = cm.prove_sharded(dm, dp)  # sizes the pools
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            pr = cm.prove_sharded(dm, dp)
        res["ms_per_proof"] = (time.perf_counter() - t0) / 3 * 1e3
        res["sha256"] = hashlib.sha256(pr.words.tobytes()).hexdigest()
        if rank == 0:
            res["same_words_as_the_single_gpu_proof"] = bool(np.array_equal(pr.words, prover.prove(dm, dp).words))
    except Exception as e:  # noqa: BLE001
        res["error"] = "%s: %s" % (type(e).__name__, e)
    print("sharded_leg rank %d of %d: %s" % (rank, world, json.dumps(res)), file=sys.stderr, flush=True)
    os._exit(0)


def spawn_ranks(n):
    """`python bench.py --gpus N` started the way the N = 1 line is started (no launcher, no WORLD_SIZE): this process becomes the launcher —
    N copies of this very command line, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), rank 0's
    stdout (the ONE JSON line) passed through, the first failing rank's status returned after the others were ended.  The
    `python -m torch.distributed.run ... bench.py --gpus N` form sets WORLD_SIZE itself and never comes here."""
    import signal
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True))
    status, deadline = 0, time.time() + float(os.environ.get("VGPU_BENCH_SPAWN_TIMEOUT", "3000"))
    alive = list(procs)
    while alive and status == 0:
        time.sleep(0.2)
        for p_ in list(alive):
            rc_ = p_.poll()
            if rc_ is not None:
                alive.remove(p_)
                if rc_ != 0:
                    status = rc_ if rc_ > 0 else 128 - rc_
                    print("bench.py: rank %d ended with status %d; ending the other ranks" % (procs.index(p_), rc_), file=sys.stderr, flush=True)
        if time.time() > deadline:
            status = 124
            print("bench.py: ranks still running at the launcher's deadline; ending them", file=sys.stderr, flush=True)
    for p_ in alive:  # exact process groups this launcher started, never a pattern
        try:
            os.killpg(p_.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
    for p_ in alive:
        try:
            p_.wait(10)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p_.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
    sys.exit(status)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--log-rows", type=int, default=20, help="log2 of the padded CPU-chip height (20 = the headline workload)")
    ap.add_argument("--cpu-log-rows", type=int, default=None, help="size of the CPU-baseline proof (default: the workload's own size, capped at 2^20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scalar-baseline", action="store_true", help="also time the oracle's scalar mode in this run (about a minute more); nothing is reported about it otherwise")
    ap.add_argument("--inflight", type=int, default=3,
                    help="proofs in flight per GPU: M prover contexts (own HIP streams and pool), one asynchronous proof outstanding on each "
                         "(one proof's latency-bound FRI/Merkle-top tail overlaps the next proof's throughput-bound commits)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the latency / PCIe-inclusive / operation-log legs (profiling runs: the process then consists of the warmup and the timed region only)")
    ap.add_argument("--sustained-seconds", type=float, default=6.0,
                    help="length of the SUSTAINED region run right after the contract's K steps (same loop, same fences; reported as `sustained`, never as `value`); 0 = skip")
    ap.add_argument("--no-clock-probe", action="store_true", help="do not sample the shader clock over the timed region (A/B of the probe itself)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="experiment: no per-launch HIP events in the timed region (then no per-kernel figures / roofline in the line)")
    ap.add_argument("--mmcs", choices=["keccak", "poseidon"], default="keccak",
                    help="Merkle hash: keccak = the reference's configuration (headline); poseidon = BASELINE.json's north-star variant "
                         "(PaddingFreeSponge / TruncatedPermutation over Poseidon-16), a separate leg with its own roofline")
    ap.add_argument("--workload", choices=["c2", "c3", "c4"], default="c2",
                    help="SURVEY.md §8 config: c2 = Fibonacci (headline), c3 = Fibonacci 2^22 rows with 4x blowup, c4 = ALU/range-heavy loop")
    ap.add_argument("--sharded-leg", action="store_true", help=argparse.SUPPRESS)  # internal: the one-proof-over-all-ranks leg of an N > 1 run, in a process of its own
    args = ap.parse_args()
    if args.sharded_leg:
        return sharded_leg_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)

    # Pin the measured configuration before the HIP runtime starts: 3 prover contexts x (main + 1 aux stream) over FOUR hardware
    # queues (the runtime's default; other counts measured worse, DESIGN.md "Measurement").
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
    # a lone rank has host cores to spare: its proof threads poll for the GPU instead of parking (libvgpu: VGPU_SPIN_WAIT); with one
    # rank per GPU the node's ranks share the cores and park
    os.environ.setdefault("VGPU_SPIN_WAIT", "1" if int(os.environ.get("WORLD_SIZE", "1")) == 1 else "0")
    import torch
    import valida_amd as va

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:  # a launcher's WORLD_SIZE wins: the line's n_gpus is the number of ranks that ran
        print("bench.py: --gpus %d but WORLD_SIZE=%d; running with %d ranks" % (args.gpus, world, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    if os.environ.get("VGPU_BENCH_TEST_EXIT_RANK") == str(rank) and world > 1:  # test hook: this rank dies before the rendezvous (tests/test_gpu_parity.py)
        sys.exit(7)
    # test hooks (a 1-GPU box exercising the N > 1 control flow): every rank on one device, gloo instead of RCCL
    backend = os.environ.get("VGPU_BENCH_BACKEND", "nccl")
    if "VGPU_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["VGPU_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    coll_device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=coll_device)
        else:
            dist.init_process_group(backend=backend)
    # Control plane (the communicator id, the agreement on the route) on a gloo group of its own: if the library's RCCL rendezvous gets
    # stuck, nothing the job still has to say to itself may travel through RCCL on the same device — the abandoned rendezvous could
    # interfere with it or deadlock it — so after a stuck bring-up EVERY collective of the job (roots, barriers, the final MAX) uses this
    # group and host tensors, and the line says so ("comm_stuck").
    ctl = None
    if world > 1 and backend == "nccl":
        try:
            ctl = dist.new_group(backend="gloo")
        except Exception as e:  # noqa: BLE001 - without it the control plane stays on the default group, as before round 3
            print("bench: no gloo control group (%s: %s); control messages use the default group" % (type(e).__name__, e), file=sys.stderr, flush=True)
    coll_group = None

    rc = va.poseidon_round_constants()
    machine = va.Machine.basic()
    if args.workload == "c3":
        args.log_rows = 22
    hash_kind = va.HASH_POSEIDON16 if args.mmcs == "poseidon" else va.HASH_KECCAK256
    provers = [va.Prover(machine, rc, log_blowup=2 if args.workload == "c3" else 1, device=local_rank, hash_kind=hash_kind) for _ in range(max(1, args.inflight))]
    prover = provers[0]

    # one independent segment per rank: distinct loop bounds with the same padded shape
    t0 = time.time()
    if args.workload == "c4":
        n = ((1 << args.log_rows) - 8) // 9 - rank  # 4 + 9n + 1 cycles
        wl = va.Workload.alu(n)
        wl_name = "C4: ALU/range-heavy loop (add, sub, xor, and, or, lt, addi, addi, bne), 2^%d cpu rows, mem 2^%d" % (args.log_rows, args.log_rows + 2)
        data = "alu_program(iters=%d)" % n
    else:
        n = segment_loop_bound(args.log_rows, rank)
        wl = va.Workload.fib(n)
        wl_name = "%s: Fibonacci 2^%d cpu rows (mem 2^%d)" % (args.workload.upper(), args.log_rows, args.log_rows + 2)
        data = "fib_program(n=%d)" % n
    t_tracegen = time.time() - t0
    assert wl.cpu_height == 1 << args.log_rows
    mt = wl.main_traces()
    prep = wl.preprocessed()
    t0 = time.time()
    dmain = [prover.upload(m) for m in mt]
    dprep = [(c, prover.upload(m)) for c, m in prep]
    t_upload = time.time() - t0
    inputs = [(dmain, dprep)] + [([p.upload(m) for m in mt], [(c, p.upload(m)) for c, m in prep]) for p in provers[1:]]
    shapes = [(m.shape[0], m.shape[1], machine.chip_info(i)["interactions"]) for i, m in enumerate(mt)]
    upload_bytes = sum(m.nbytes for m in mt)

    all_roots = [None]
    # The path's one collective is issued by the LIBRARY (vgpu_comm_*: RCCL loaded and driven by libvgpu.so, so a non-Python host
    # owns it); the 128-byte communicator id travels through the launcher's own channel (here torch.distributed's).  Any failure
    # falls back to torch.distributed's all_gather and says so in the JSON line.
    lib_comm, comm_note, comm_stuck = None, "torch.distributed all_gather_into_tensor", False
    if world > 1 and backend == "nccl" and os.environ.get("VGPU_BENCH_COMM", "library") == "library":
        # a rendezvous that never completes must not take the scaling run with it: communicator creation and a first all-gather
        # run on a watchdog thread; past the deadline this rank reports failure and every rank takes the torch route
        import threading
        box = {}

        def bring_up():
            try:
                cm = va.Comm(prover, ids[0], rank, world)
                probe = cm.allgather_roots(np.full(24, rank, dtype=np.uint32))
                if [int(r[0]) for r in probe] != list(range(world)):
                    raise RuntimeError("probe all-gather returned %s" % probe[:, 0].tolist())
                # every later collective of the library is bounded: a rank that died mid-run aborts the communicator of its peers (ncclCommAbort)
                # after this long instead of hanging them until the driver's own limit
                cm.set_timeout_ms(int(os.environ.get("VGPU_BENCH_COLLECTIVE_TIMEOUT_MS", "120000")))
                box["comm"] = cm
            except Exception as e:  # noqa: BLE001 - the scaling run must not die on the optional route
                box["error"] = e

        try:
            ids = [va.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0, group=ctl)
            th = threading.Thread(target=bring_up, daemon=True)
            th.start()
            th.join(float(os.environ.get("VGPU_BENCH_COMM_TIMEOUT", "90")))
            if th.is_alive():
                comm_stuck = True
                raise TimeoutError("communicator bring-up still running after the deadline")
            if "error" in box:
                raise box["error"]
            lib_comm = box["comm"]
            comm_note = "vgpu_comm_allgather_roots (RCCL driven by libvgpu.so)"
        except Exception as e:  # noqa: BLE001
            lib_comm, comm_note = None, "torch.distributed all_gather_into_tensor (library communicator failed: %s)" % e
        gathered = [None] * world
        dist.all_gather_object(gathered, (lib_comm is not None, comm_stuck), group=ctl)
        if any(g[1] for g in gathered):  # some rank left a thread inside the rendezvous: no RCCL traffic of any kind from here on
            comm_stuck = True
            if ctl is not None:
                coll_group, coll_device = ctl, torch.device("cpu")
        if not all(g[0] for g in gathered):  # all ranks take the same route
            lib_comm = None
            if "failed" not in comm_note:
                comm_note = "torch.distributed all_gather_into_tensor (library communicator failed on another rank)"

    oplog = wl.oplog()
    small = [i for i in range(va.NUM_CHIPS) if i not in va.GENERATED_CHIPS]

    def run_steps(k, from_host=False, from_oplog=False, host_traces=None, generator=None):
        """k proofs on this GPU, then the path's one collective.  With --inflight M > 1 the steps go round-robin over M
        prover contexts through the library's asynchronous prove, so one proof's latency-bound Merkle-top / FRI tail
        overlaps another's throughput-bound commits."""
        done = [None] * k

        def generate(g):
            log = g.upload_oplog(oplog)
            tr = {c: g.generate_trace(log, c) for c in va.GENERATED_CHIPS}
            tr.update({c: g.upload(mt[c]) for c in small})
            return [tr[c] for c in range(va.NUM_CHIPS)], (log, g)

        def start(slot):
            pr = provers[slot]
            if from_oplog:  # H2D of the VM's operation logs, Chip::generate_trace on the device, small chips uploaded
                tr, keep = generate(pr)
                return pr.prove_async(tr, inputs[slot][1], keep=keep)
            if from_host:  # the boundary handing over host buffers: H2D of the 14 main traces inside the step
                return pr.prove_async([pr.upload(m) for m in (host_traces or mt)], inputs[slot][1])
            return pr.prove_async(*inputs[slot])

        # one caller thread, one outstanding ticket per prover context (vgpu_prove_async): step i runs on context i % M
        tickets = [None] * len(provers)

        def finish(j, t):
            done[j] = t.wait()
            if world > 1:  # the path's one collective, once per proof: every rank's three roots to every rank (96 B per rank)
                if lib_comm is not None:
                    all_roots[0] = lib_comm.allgather_roots(done[j].words[2:26])
                else:
                    all_roots[0] = exchange_roots(dist, torch, done[j].words[2:26], coll_device, coll_group)

        for i in range(k):
            slot = i % len(provers)
            staged = None
            if from_oplog and generator is not None:  # generate on the side context BEFORE waiting for this slot's previous proof
                tr, keep = generate(generator)
                if tickets[slot] is not None:
                    finish(*tickets[slot])
                tickets[slot] = (i, provers[slot].prove_async(tr, inputs[slot][1], keep=keep))
                continue
            if from_host:  # the next segment's matrices go up while this context still proves the previous one (the library copies
                staged = [provers[slot].upload(m) for m in (host_traces or mt)]  # beside a running proof on a stream of its own)
            if tickets[slot] is not None:
                finish(*tickets[slot])
            tickets[slot] = (i, provers[slot].prove_async(staged, inputs[slot][1]) if staged is not None else start(slot))
        for tk in sorted((t for t in tickets if t is not None), key=lambda t: t[0]):
            finish(*tk)
        return done

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group=coll_group)
        torch.cuda.synchronize()

    # context initialisation (not a step): every prover context sizes its memory pool and loads its kernels once, so the W
    # warmup steps below mean the same thing whatever the number of contexts
    for pr, inp in zip(provers, inputs):
        pr.prove(*inp)
    run_steps(args.warmup)
    # prover latency with nothing else in flight (the "prover ms" half of the metric)
    single, prof1 = None, None
    if len(provers) > 1 and not args.no_extra_legs:
        # latency of one proof alone: no per-launch HIP events (two hipEventRecord per launch cost the host ~2 ms per proof, which a
        # lone proof cannot hide), then two further proofs WITH events for this leg's per-kernel figures
        fence()
        t0 = time.perf_counter()
        for _ in range(6):
            provers[0].prove(*inputs[0])
        torch.cuda.synchronize()
        single = (time.perf_counter() - t0) / 6 * 1e3
        provers[0].set_profiling(True)
        for _ in range(2):
            provers[0].prove(*inputs[0])
        prof1 = provers[0].profile()
        provers[0].set_profiling(False)
    pcie_ms = oplog_ms = pcie_pinned_ms = sharded_local = None
    if not args.no_extra_legs:
        # the same boundary with the host matrices in page-locked memory (vgpu_host_alloc): the upload is one DMA per matrix
        mt_pinned = [va.pinned_copy(m) for m in mt]
        run_steps(len(provers), from_host=True, host_traces=mt_pinned)
        fence()
        t0 = time.perf_counter()
        run_steps(4 * len(provers), from_host=True, host_traces=mt_pinned)
        torch.cuda.synchronize()
        pcie_pinned_ms = (time.perf_counter() - t0) / (4 * len(provers)) * 1e3
        del mt_pinned
        # PCIe-inclusive rate (never `value`): every step starts from the host-resident main traces
        fence()
        t0 = time.perf_counter()
        run_steps(2 * len(provers), from_host=True)
        torch.cuda.synchronize()
        pcie_ms = (time.perf_counter() - t0) / (2 * len(provers)) * 1e3
        # the same with device trace generation: every step starts from the host-resident operation logs
        gen = va.Prover(machine, rc, log_blowup=2 if args.workload == "c3" else 1, device=local_rank, hash_kind=hash_kind) if len(provers) > 1 else None
        run_steps(len(provers), from_oplog=True, generator=gen)
        fence()
        t0 = time.perf_counter()
        run_steps(4 * len(provers), from_oplog=True, generator=gen)
        torch.cuda.synchronize()
        oplog_ms = (time.perf_counter() - t0) / (4 * len(provers)) * 1e3
        del gen
        # ONE proof over W prover contexts of this GPU standing in for W ranks (vgpu_prove_sharded_local).  All the work
        # still runs on this one device, so the figures show what sharding COSTS (exchanges as device-to-device copies, the small chips and
        # the permutation traces computed by every rank, the host steps between the phases) — not a speed-up.
        if world == 1 and os.environ.get("VGPU_BENCH_SHARDED", "1") == "1":
            sharded_local = {}
            ref_words = provers[0].prove(*inputs[0]).words
            extra = []
            big = args.workload == "c3"  # 41 GB of pool per single-GPU context: the leg's contexts share the 288 GB with them
            try:
                for wn in ((1, 8) if big else (1, 2, 4, 8)):
                    if big:
                        for p_ in provers + extra:
                            p_.trim()  # pools of the other shapes back to the driver first
                    while len(provers) + len(extra) < wn:
                        extra.append(va.Prover(machine, rc, log_blowup=prover.log_blowup, device=local_rank, hash_kind=hash_kind))
                    ps = (provers + extra)[:wn]
                    up = va.upload_replicated(ps, mt, prep)
                    pr = va.prove_sharded_local(ps, mt, prep, uploaded=up)  # sizes the pools
                    same = bool(np.array_equal(pr.words, ref_words))
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        va.prove_sharded_local(ps, mt, prep, uploaded=up)
                    torch.cuda.synchronize()
                    sharded_local[str(wn)] = {"ms_per_proof": (time.perf_counter() - t0) / 3 * 1e3, "same_words_as_the_single_gpu_proof": same}
                    del up
                    if wn > 1:  # the same with the TRACES sharded too (row-range inputs: every context holds 1 / W of the big chips' rows)
                        up = va.upload_row_ranges(ps, mt, prep)
                        pr = va.prove_sharded_rows_local(ps, mt, prep, uploaded=up)
                        same = bool(np.array_equal(pr.words, ref_words))
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(3):
                            va.prove_sharded_rows_local(ps, mt, prep, uploaded=up)
                        torch.cuda.synchronize()
                        sharded_local[str(wn)]["row_range_inputs"] = {"ms_per_proof": (time.perf_counter() - t0) / 3 * 1e3, "same_words_as_the_single_gpu_proof": same}
                        del up
            except va.VgpuError as e:  # reporting leg only: never lose the line over it
                sharded_local["error"] = str(e)
            del extra
            if big:
                for p_ in provers:
                    p_.trim()
                run_steps(len(provers))  # the pools size themselves again (first-touch hipMalloc inside the table's steps would be timed as kernel time)
                torch.cuda.synchronize()
    # Per-kernel table: a few steps with events on EVERY launch, outside the timed region.  Timed region: events on the launches
    # of the dominant kernel only — a timed launch carries a pair of events, and 1270 of them per proof cost ~4 % of the throughput
    # being measured; the roofline needs that kernel's live average, the table does not need the timed region.
    pool_peak_whole = sum(p_.memory()[1] for p_ in provers)  # with the legs above (uploads of the PCIe legs, the sharded leg's shards) in it
    for p_ in provers:
        p_.reset_memory_peak()  # from here on: the proving path alone (kernel table, contract region, sustained region)
        p_.set_profiling(True)
    run_steps(2 * len(provers))
    torch.cuda.synchronize()
    table = {}
    for p_ in provers:
        for k, v in p_.profile().items():
            a = table.get(k, (0, 0.0, 0.0, 0.0))
            table[k] = (a[0] + v[0], a[1] + v[1], a[2] + v[2], a[3] + v[3])
    table_steps = 2 * len(provers)
    # The dominant kernel: by accumulated duration with ONE proof on the GPU when that leg ran (exclusive times) — with several proofs in
    # flight a small kernel starved of wave slots by the others' big launches (C3: k_bary_weights behind 10 ms NTT launches) accumulates
    # the longest begin-to-end spans without doing the most work
    dominant = max((prof1 or table).items(), key=lambda kv: kv[1][1])[0]
    if dominant not in table:
        dominant = max(table.items(), key=lambda kv: kv[1][1])[0]
    # The shader clock the device sustains under THIS load, sampled over a timed region by one probing wave every 40 ms (libvgpu:
    # vgpu_shader_clock_probe; rocm-smi is blind here): the rooflines' issue bounds are priced at the guide's 2.4 GHz AND at this clock.
    import threading

    if not args.no_clock_probe:
        va.shader_clock_hz(local_rank, 4096)  # creates the probe's stream outside the timed regions

    def timed_region(k):
        """EXACTLY k steps between two fences; returns (seconds, phase sums, per-kernel profile of the region, clock samples, last proof)."""
        samples, stop, th = [], None, None
        if not args.no_clock_probe:
            stop = threading.Event()

            def sample_clock():
                try:
                    while not stop.is_set():
                        samples.append(va.shader_clock_hz(local_rank, 2048))
                        stop.wait(0.04)
                except Exception as e:  # noqa: BLE001 - a measurement aid must not cost the line
                    print("bench: clock probe stopped (%s: %s)" % (type(e).__name__, e), file=sys.stderr, flush=True)

            th = threading.Thread(target=sample_clock, daemon=True)
        for p_ in provers:  # events on the dominant kernel's launches only; resets the accumulators
            p_.set_profiling(not args.no_kernel_events, only=dominant)
        fence()
        if th is not None:
            th.start()
        t0_ = time.perf_counter()
        phase_, last = {}, None
        for last in run_steps(k):
            for k_, v in last.phase_ms.items():
                phase_[k_] = phase_.get(k_, 0.0) + v
        fence()
        dt = time.perf_counter() - t0_
        if stop is not None:
            stop.set()
            th.join(5)
        prof_ = {}
        for p_ in provers:
            for k_, v in p_.profile().items():
                a = prof_.get(k_, (0, 0.0, 0.0, 0.0))
                prof_[k_] = (a[0] + v[0], a[1] + v[1], a[2] + v[2], a[3] + v[3])
        return dt, phase_, prof_, samples, last

    elapsed, phase, prof, clock_samples, p = timed_region(args.steps)  # THE contract region: W warmup steps done above, exactly K steps here
    # The SUSTAINED figure: the same loop for >= --sustained-seconds right behind the contract region (round-5 verdict: a 0.3 s burst runs at
    # 2.35-2.4 GHz, the device settles at 2.2 GHz after a few seconds of this load; a production prover lives in the second regime).
    sustained = None
    if args.sustained_seconds > 0:
        s_steps = max(args.steps, int(np.ceil(1.15 * args.sustained_seconds / (elapsed / args.steps))))  # a count, not a deadline: the ranks must agree on it (one collective per step)
        s_steps = -(-s_steps // len(provers)) * len(provers)
        if world > 1:  # every rank must run the same number of steps (one collective per step)
            ns = torch.tensor([s_steps], dtype=torch.int64, device=coll_device)
            dist.all_reduce(ns, op=dist.ReduceOp.MAX, group=coll_group)
            s_steps = int(ns.item())
        s_elapsed, _, s_prof, s_clock, _ = timed_region(s_steps)
        if world > 1:
            tm = torch.tensor([s_elapsed], dtype=torch.float64, device=coll_device)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX, group=coll_group)
            s_elapsed = float(tm.item())
        sustained = {"steps": s_steps, "seconds": s_elapsed, "prof": s_prof, "clock": s_clock}
    for p_ in provers:
        p_.set_profiling(False)

    # the last proof of the timed region through the library's own Machine::verify (host code; never inside a timed region)
    verified = None
    try:
        t0 = time.perf_counter()
        pc_host = va.host_commit_root([m for _, m in prep], rc, log_blowup=prover.log_blowup, hash_kind=hash_kind)
        msg = va.verify(machine, rc, p.words, pc_host, log_blowup=prover.log_blowup, num_queries=prover.num_queries, pow_bits=prover.pow_bits, hash_kind=hash_kind)
        verified = {"accepted": msg is None, "ms": (time.perf_counter() - t0) * 1e3}
        if msg is not None:
            verified["reason"] = msg
    except Exception as e:  # noqa: BLE001 - reporting only
        verified = {"accepted": False, "reason": "%s: %s" % (type(e).__name__, e)}

    rank_devices = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=coll_group)
        elapsed = float(tmax.item())
        # which device every rank ran on (the line must show one rank per GPU, or say that it was a one-device stand-in)
        try:
            props = torch.cuda.get_device_properties(local_rank)
            mine = {"rank": rank, "device": local_rank, "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", ""))}
        except Exception as e:  # noqa: BLE001 - reporting only
            mine = {"rank": rank, "device": local_rank, "error": str(e)}
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine, group=ctl if ctl is not None else coll_group)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        if not prof:  # --no-kernel-events experiment: no per-kernel figures, hence no roofline: not a contract line
            print(json.dumps({"metric": "proofs/sec (experiment without per-launch events)", "value": value, "unit": "proofs/s", "ms_per_step": ms_per_step,
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prover_ms_single_proof_in_flight": single}))
            return
        # the dominant kernel chosen above (the only one timed in the region)
        name = dominant if dominant in prof else max(prof.items(), key=lambda kv: kv[1][1])[0]
        launches, ms, nbytes, valu_ops = prof[name]
        achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        valu_rate = valu_ops / (ms * 1e-3) if ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic_per_launch(name)
        mb = microbench_facts()
        simds = 1024
        # Keccak kernels: permutations per second against (a) the product's own permutation running in registers with nothing else
        # (tools/microbench: the ceiling this code can reach) and (b) the sum of its instructions at their isolated issue rates
        def keccak_roofline(stat):
            """VALU roofline of a Merkle kernel: permutations per second against the ISSUE BOUND of the permutation's instruction mix (the
            guide's 2 / 4 cycles per full- / half-rate wave64 instruction at 2.4 GHz on 1024 SIMDs).  Beside it, labelled as what they are:
            the same mix at the MEASURED isolated rates, and the code's in-register ceiling (a sample of what this very code reaches with
            no memory traffic — NOT a bound: a launch can exceed it)."""
            launches_, ms_, _, valu_ = stat
            if not valu_ or not ms_:
                return None
            if name.startswith("k_poseidon"):  # Poseidon-16 kernels: Montgomery products, i.e. half-rate multiply instructions
                full_pp, half_pp, unit = POSEIDON_FULL_PER_PERM, POSEIDON_HALF_PER_PERM, "Poseidon-16 permutations/s"
                mix = "%d half-rate (multiplies) + %d full-rate VALU instructions per permutation as the kernels run it (sparse partial rounds, MDS layer as CRT blocks)" % (half_pp, full_pp)
                ceiling = mb.get("poseidon_perm_per_s")
            else:
                full_pp, half_pp, unit = 24 * KECCAK_FULL_RATE_PER_ROUND, 24 * KECCAK_HALF_RATE_PER_ROUND, "Keccak-f[1600] permutations/s"
                mix = "24 rounds x (122 full-rate + 56 half-rate VALU instructions)"
                ceiling = mb.get("keccak_perm_per_s")
            perms = valu_ * 64.0 / (POSEIDON_HALF_PER_PERM + POSEIDON_FULL_PER_PERM if name.startswith("k_poseidon") else KECCAK_VALU_PER_PERM)
            rate = perms / (ms_ * 1e-3)
            bound = issue_bound_per_s(full_pp, half_pp)
            measured = simds * mb["clock_hz"] * 64.0 / (full_pp * simds * mb["clock_hz"] / mb["full_rate"] + half_pp * simds * mb["clock_hz"] / mb["half_rate"])
            out_ = {"achieved": rate, "unit": unit, "peak": bound, "frac": rate / bound,
                    "peak_is": "issue bound: %s at 2 / 4 SIMD-cycles per full- / half-rate wave64 instruction, 1024 SIMDs, 2.4 GHz (MI355X_MICROARCH.md)" % mix,
                    "measured_issue_peak": measured, "frac_of_measured_issue_peak": rate / measured,
                    "measured_issue_peak_is": "the same mix at the isolated per-class rates measured on this chip (%s)" % mb["source"],
                    "permutations_per_launch": perms / launches_ if launches_ else None}
            if ceiling:
                out_["code_ceiling"] = ceiling
                out_["frac_of_code_ceiling"] = rate / ceiling
                out_["code_ceiling_is"] = "NOT a bound: the product's own permutation chained in registers with no memory traffic (tools/microbench.hip), a sample of what this code reaches"
            assert out_["frac"] <= 1.0, "a roofline fraction above 1: the peak is not a bound (%r)" % (out_,)
            return out_
        # whole proof: VALU wave-instructions per proof from the committed PMC pass x this run's step time
        _, pmc_kernels = pmc_file()
        proof_instr = None
        try:
            proof_instr = sum(pmc_kernels[k]["valu_wave_instr_per_launch"] * v[0] / table_steps for k, v in table.items() if k in pmc_kernels and "valu_wave_instr_per_launch" in pmc_kernels[k])
        except (KeyError, TypeError):
            proof_instr = None
        b_alg = algorithmic_bytes_per_proof(shapes, prover.log_blowup)
        out = {
            # BASELINE.json: "proofs/sec + prover ms, 2^20-row Fibonacci trace": `value` is the proofs/sec half, the prover ms
            # half is `prover_ms_single_proof_in_flight`
            "metric": "proofs/sec, 2^%d-row %s trace" % (args.log_rows, "ALU-loop" if args.workload == "c4" else "Fibonacci"),
            "value": value,
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",  # BabyBear mod p in Montgomery form; Keccak lanes as pairs of u32
            "data": "synthetic: %s traces generated by the in-tree VM, deterministic; Poseidon constants SplitMix64(0x56414C494441); every step proves the SAME "
                    "device-resident trace from scratch (nothing is cached between proofs); ranks differ by their loop bound" % data,
            "config": {
                "workload": wl_name + ", 14 chips, one proof per GPU",
                "field": "BabyBear / Ext5", "mmcs": "Keccak-256" if args.mmcs == "keccak" else "Poseidon-16 (PaddingFreeSponge<16,8,8> / TruncatedPermutation<2,8,16>; north-star variant, not the reference's)", "fri_blowup": 1 << prover.log_blowup,
                "log_blowup": prover.log_blowup, "num_queries": prover.num_queries, "pow_bits": prover.pow_bits,
                "parallelism": ("segments: one independent proof per GPU + all-gather of 3 roots per proof via " + comm_note) if world > 1 else "single GPU",
                "proofs_in_flight_per_gpu": len(provers),
                "comm_stuck": bool(comm_stuck),  # true: the library's RCCL bring-up timed out somewhere and the job's collectives ran over gloo
                # N > 1: what the collective saw in the LAST step — rows of the gathered root table (one per rank), how many of them differ
                # (segments differ by loop bound, so every rank's roots are its own), and the device each rank ran on
                "collective": None if world == 1 else {
                    "backend": backend, "route": comm_note, "ranks_in_last_allgather": None if all_roots[0] is None else int(all_roots[0].shape[0]),
                    "distinct_root_sets": None if all_roots[0] is None else len({tuple(int(x) for x in r) for r in all_roots[0]}),
                    "rank_devices": rank_devices,
                    "distinct_devices": None if not rank_devices else len({(d.get("device"), d.get("uuid")) for d in rank_devices})},
            },
            "roofline": roofline_object(name, prof[name], achieved, traffic, traffic_src, keccak_roofline(prof[name]) if valu_ops > 0 else None, args.steps),
            # The dominant kernel is Keccak-f[1600] over Merkle nodes: 32-bit integer VALU work, ~4200 instructions per 96 B moved, so
            # its binding roofline is the VALU issue rate, not HBM (SURVEY.md §8(d) caveat; DESIGN.md "Rooflines").
            "valu_roofline": None if valu_ops <= 0 else dict(keccak_roofline(prof[name]), kernel=name, microbench=mb["source"],
                                                            wave64_valu_instr_per_s=valu_rate, full_rate_peak=mb["full_rate"], half_rate_peak=mb["half_rate"]),
            # The same kernel when its proof has the GPU to itself (the 4-proof latency leg): with several proofs in flight kernels
            # of all of them share the CUs, so the per-launch durations above are not exclusive-use figures.
            "roofline_one_proof_in_flight": None if not prof1 else {
                "kernel": name, "avg_launch_ms": prof1[name][1] / prof1[name][0],
                "hbm_achieved_GBs": prof1[name][2] / (prof1[name][1] * 1e-3) / 1e9, "hbm_frac": prof1[name][2] / (prof1[name][1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "valu": keccak_roofline(prof1[name]),
            },
            "proof_valu_roofline": None if not proof_instr else {
                "wave64_valu_instr_per_proof": proof_instr, "achieved_instr_per_s": proof_instr / (ms_per_step * 1e-3),
                "issue_bound_full_rate": GUIDE_SIMDS * GUIDE_CLOCK_HZ / GUIDE_FULL_CYCLES,
                "frac_of_issue_bound_full_rate": proof_instr / (ms_per_step * 1e-3) / (GUIDE_SIMDS * GUIDE_CLOCK_HZ / GUIDE_FULL_CYCLES),
                "frac_of_issue_bound_half_rate": proof_instr / (ms_per_step * 1e-3) / (GUIDE_SIMDS * GUIDE_CLOCK_HZ / GUIDE_HALF_CYCLES),
                "frac_of_issue_bound_full_rate_at_measured_clock": None if not clock_samples else proof_instr / (ms_per_step * 1e-3) / (GUIDE_SIMDS * (sum(clock_samples) / len(clock_samples)) / GUIDE_FULL_CYCLES),
                "sustained_frac_of_issue_bound_full_rate": None if not sustained else proof_instr / (sustained["seconds"] / sustained["steps"]) / (GUIDE_SIMDS * GUIDE_CLOCK_HZ / GUIDE_FULL_CYCLES),
                "sustained_frac_of_issue_bound_full_rate_at_measured_clock": None if not sustained or not sustained["clock"] else
                    proof_instr / (sustained["seconds"] / sustained["steps"]) / (GUIDE_SIMDS * (sum(sustained["clock"]) / len(sustained["clock"])) / GUIDE_FULL_CYCLES),
                "frac_of_full_rate_peak": proof_instr / (ms_per_step * 1e-3) / mb["full_rate"], "frac_of_half_rate_peak": proof_instr / (ms_per_step * 1e-3) / mb["half_rate"],
                "note": "instruction counts from the committed PMC pass (SQ_INSTS_VALU per launch) x this run's launches per step; the proof's mix of full- and "
                        "half-rate instructions puts its issue peak between the two"},
            "shader_clock": None if not clock_samples else {
                "GHz_mean": sum(clock_samples) / len(clock_samples) / 1e9, "GHz_min": min(clock_samples) / 1e9, "GHz_max": max(clock_samples) / 1e9, "samples": len(clock_samples),
                "guide_GHz": GUIDE_CLOCK_HZ / 1e9,
                "roofline_frac_at_the_measured_clock": None if valu_ops <= 0 else keccak_roofline(prof[name])["frac"] * GUIDE_CLOCK_HZ / (sum(clock_samples) / len(clock_samples)),
                "how": "one wave of another stream every 40 ms over the timed region: shader cycles of 2048 dependent VALU additions against the 100 MHz wall clock "
                       "(vgpu_shader_clock_probe); the issue bounds in `roofline` / `valu_roofline` / `proof_valu_roofline` are priced at the guide's clock"},
            # >= --sustained-seconds of the same loop right behind the contract region: what a prover that never stops delivers (the device's clock
            # management settles below the burst clock); the dominant kernel's roofline of THAT region priced at the guide's clock and at its own
            "sustained": None if not sustained else sustained_object(sustained, world, name, keccak_roofline, b_alg),
            "proof_roofline": {"algorithmic_bytes_per_proof": b_alg, "achieved_GBs": b_alg / (ms_per_step * 1e-3) / 1e9, "frac_of_hbm_peak": b_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "phase_ms": {k: v / args.steps for k, v in phase.items()},
            "kernel_ms_per_step": {k: v[1] / table_steps for k, v in sorted(table.items(), key=lambda kv: -kv[1][1])},
            # the same table with ONE proof on the GPU (exclusive spans: with several proofs in flight the spans of their kernels overlap and a
            # span / bytes figure of a small kernel is not interpretable — C3's table must be read here)
            "kernel_ms_per_step_one_proof_in_flight": None if not prof1 else {k: v[1] / 2 for k, v in sorted(prof1.items(), key=lambda kv: -kv[1][1])},
            "kernel_GBs_one_proof_in_flight": None if not prof1 else {k: (v[2] / (v[1] * 1e-3) / 1e9 if v[1] > 0 else 0.0) for k, v in prof1.items()},
            "kernel_GBs": {k: (v[2] / (v[1] * 1e-3) / 1e9 if v[1] > 0 else 0.0) for k, v in table.items()},
            "kernel_ms_total_per_step": sum(v[1] for v in table.values()) / table_steps,
            "kernel_table_source": "%d steps with events on every launch, run just before the timed region (same configuration); the timed region times "
                                   "the dominant kernel (%s) only" % (table_steps, dominant),
            "outside_timed_region": {"tracegen_s": t_tracegen, "h2d_upload_s": t_upload, "h2d_bytes": upload_bytes},
            "pcie_inclusive": None if pcie_ms is None else {
                "ms_per_step": pcie_ms, "value": world * 1e3 / pcie_ms, "unit": "proofs/s", "h2d_bytes_per_step": upload_bytes,
                "note": "main traces uploaded from pageable host memory for every step, staged while the context proves the previous segment; not the headline value",
                "from_pinned_host_memory": None if pcie_pinned_ms is None else {
                    "ms_per_step": pcie_pinned_ms, "value": world * 1e3 / pcie_pinned_ms,
                    "note": "the same with the host matrices allocated by vgpu_host_alloc (page-locked): one DMA per matrix"}},
            "from_operation_logs": None if oplog_ms is None else {"ms_per_step": oplog_ms, "value": world * 1e3 / oplog_ms, "unit": "proofs/s",
                                    "h2d_bytes_per_step": int(48 * oplog.n_cpu + 16 * oplog.n_mem + 16 * sum(oplog.n_alu) + sum(mt[c].nbytes for c in small)),
                                    "note": "operation logs uploaded and all 14 chip traces generated on the device for every step, on a prover context of their own "
                                            "while the proving contexts work on the previous segments (replaces host generate_trace, reported as "
                                            "tracegen_s, and the row-major upload)"},
            "one_proof_over_w_ranks_on_this_gpu": None if not sharded_local else dict(sharded_local, note=(
                "vgpu_prove_sharded_local: W prover contexts of this ONE device stand in for W ranks (row-range shards, exchanges as device-to-device "
                "copies): what sharding costs, not a speed-up; row_range_inputs = vgpu_prove_sharded_rows_local, the traces themselves sharded; the multi-GPU realisation (vgpu_prove_sharded over RCCL) is unmeasured here")),
            "prover_ms_single_proof_in_flight": single if single is not None else ms_per_step,
            "proof_words": int(p.words.size),
            "proof_sha256": hashlib.sha256(p.bytes()).hexdigest(),  # the LAST proof of the timed region (the one vgpu_verify checks below)
            "proof_checked_by_vgpu_verify": verified,  # Machine::verify of the library (host) on the last timed proof
            # peak of the contexts' HBM pools over the PROVING path (kernel table, contract and sustained regions); the reporting legs before it (PCIe uploads,
            # operation logs, one proof over W contexts) have their own, larger, high-water mark
            "hbm_pool_peak_bytes": sum(p_.memory()[1] for p_ in provers),
            "hbm_pool_peak_bytes_with_reporting_legs": max(pool_peak_whole, sum(p_.memory()[1] for p_ in provers)),
        }
        if not args.no_cpu_baseline and world == 1:
            headline = min(args.log_rows, 20)
            out["cpu_baseline"] = cb = cpu_baseline(args.cpu_log_rows if args.cpu_log_rows else headline, rc, args.log_rows, mmcs_poseidon=args.mmcs == "poseidon",
                                                    scalar_too=args.scalar_baseline)
            # north_star: "bit-identical proof bytes ... in the same run" — the CPU leg proved rank 0's own segment (same loop bound, same constants), so
            # the two hashes are over the same statement; null when the CPU sample is a smaller segment than the workload
            out["same_proof_as_cpu_baseline"] = (cb["proof_sha256"] == out["proof_sha256"]) if "note" not in cb and args.workload == "c2" else None
            if "note" not in cb and args.workload == "c2":  # the baseline proved the very workload `value` is quoted on
                # BASELINE.md publishes no number; its one quantitative target is ">= 20x the CPU baseline timed in the same run" (section 2):
                # that ratio, against what the baseline IS here (kind "port": scalar restatement, not Plonky3's packed AVX prover)
                out["vs_baseline"] = value / cb["value"]
                out["vs_baseline_is"] = "value / cpu_baseline.value of this run (kind: %s — a tuned restatement, not Plonky3 itself; BASELINE.md section 2 target >= 20x; no published number exists)" % cb["kind"]
        print(json.dumps(out), flush=True)
    if world > 1 and lib_comm is not None and not comm_stuck and os.environ.get("VGPU_BENCH_SHARDED", "1") == "1":
        # After the contract line (nothing below can change it): ONE proof — rank 0's segment — over ALL the ranks (vgpu_prove_sharded over
        # RCCL / xGMI: row-range shards), reported on stderr.  It is the first run of that path on real multi-GPU
        # hardware, so every rank runs it in a PROCESS OF ITS OWN (`bench.py --sharded-leg`, its own rendezvous on MASTER_PORT + 17, its own
        # communicator): a crash or a hang there is killed at the deadline and cannot touch this process's exit status.
        import subprocess
        dist.barrier(group=coll_group)
        env = dict(os.environ, VGPU_SHARDED_LEG_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 17))
        cmd = [sys.executable, os.path.abspath(__file__), "--sharded-leg", "--log-rows", str(args.log_rows), "--workload", args.workload, "--mmcs", args.mmcs]
        try:
            r = subprocess.run(cmd, env=env, timeout=float(os.environ.get("VGPU_BENCH_SHARDED_TIMEOUT", "240")), stdout=subprocess.DEVNULL)
            if r.returncode != 0:
                print("sharded_leg rank %d of %d: the leg's process ended with status %d" % (rank, world, r.returncode), file=sys.stderr, flush=True)
        except subprocess.TimeoutExpired:
            print("sharded_leg rank %d of %d: still running after the deadline; killed" % (rank, world), file=sys.stderr, flush=True)
        except Exception as e:  # noqa: BLE001 - an optional leg
            print("sharded_leg rank %d of %d: %s: %s" % (rank, world, type(e).__name__, e), file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(0)
    if world > 1:
        if comm_stuck:  # a thread is still inside the abandoned rendezvous: leave without waiting for it
            dist.barrier(group=coll_group)
            sys.stdout.flush()
            os._exit(0)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
