#!/usr/bin/env python3
"""Turn the rocprofv3 (rocpd SQLite) outputs of tools/profile_round.sh into the small tracked summaries under
profiles/:   python tools/summarize_prof.py r01 [outdir]
  profiles/<tag>_kernel_stats_{inflight1,default}.csv   per-kernel calls / total / average duration (us)
  profiles/<tag>_pmc.json                          per-kernel counters per launch: VALU instructions, issue
                                                   utilisation, HBM bytes (FETCH_SIZE / WRITE_SIZE, corrected as
                                                   /opt/skills/guides/MI355X_MICROARCH.md prescribes)
  profiles/<tag>_bench_*.json                      the bench lines of the same session
"""
import glob
import json
import os
import re
import shutil
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# device kernels that the library's profiler (and bench.py's kernel table) files under ANOTHER kernel's name: variants of one launch site
ALIASES = {"k_reduce_openings_rows": "k_reduce_openings", "k_perm_recip_native": "k_perm_recip"}


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    n = m.group(1) if m else name.split("(")[0]
    return ALIASES.get(n, n)


def db_of(d):
    hits = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return hits[0] if hits else None


def kernel_stats(db_path, out_csv, header):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out_csv, "w") as f:
        f.write("# %s\n# from the rocpd database view top_kernels; durations in microseconds\n" % header)
        f.write("name,total_calls,total_duration_us,average_us,percentage\n")
        for n, c, t, a, p in rows:
            f.write('"%s",%d,%.3f,%.3f,%.3f\n' % (n, c, t, a, p))
    return {short(n): (c, t, a) for n, c, t, a, p in rows}


def timeline(db_path, out_txt, header):
    """One proof alone on the GPU: where the wall time of each phase goes — busy (union of kernel intervals), idle gaps between
    kernels, and the launches, per kernel class, of the LAST proof in the trace (delimited by k_ingest launches)."""
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        return
    starts = [i for i, r in enumerate(rows) if "k_ingest" in r[0] and (i == 0 or "k_ingest" not in rows[i - 1][0])]
    if len(starts) < 2:
        return
    a, b = starts[-2], starts[-1]  # a complete proof
    seg = rows[a:b]
    t0, t1 = seg[0][1], max(r[2] for r in seg)
    busy, cur_s, cur_e = 0, seg[0][1], seg[0][2]
    for _, s_, e_ in seg[1:]:
        if s_ > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    busy += cur_e - cur_s
    per = {}
    for n, s_, e_ in seg:
        k = per.setdefault(short(n), [0, 0])
        k[0] += 1
        k[1] += e_ - s_
    with open(out_txt, "w") as f:
        f.write("# %s\n# one complete proof (kernels between two consecutive k_ingest groups), times in microseconds\n" % header)
        f.write("span_us %.1f  busy_us %.1f  idle_gaps_us %.1f  launches %d\n" % ((t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3, len(seg)))
        # gaps by size
        gaps = []
        prev_end = seg[0][2]
        for _, s_, e_ in seg[1:]:
            if s_ > prev_end:
                gaps.append(s_ - prev_end)
            prev_end = max(prev_end, e_)
        for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 20e3), (20e3, 1e12)):
            g = [x for x in gaps if lo <= x < hi]
            f.write("gaps %5.0f-%-8s us: %4d  total %.1f us\n" % (lo / 1e3, "%.0f" % (hi / 1e3) if hi < 1e11 else "inf", len(g), sum(g) / 1e3))
        # the largest idle gaps and the kernels either side (host synchronisation points show up here)
        big, prev_end, prev_name = [], seg[0][2], short(seg[0][0])
        for n_, s_, e_ in seg[1:]:
            if s_ > prev_end:
                big.append((s_ - prev_end, prev_name, short(n_), (s_ - t0) / 1e3))
            if e_ >= prev_end:
                prev_end, prev_name = e_, short(n_)
        f.write("largest gaps: gap_us  after_kernel -> before_kernel  (at_us from the start of the proof)\n")
        for g_, a_, b_, at_ in sorted(big, reverse=True)[:40]:
            f.write("  %8.1f  %-22s -> %-22s  (%.0f)\n" % (g_ / 1e3, a_, b_, at_))
        # exclusive time: how long a kernel class ran with NO other kernel on the GPU — what shortening that class would take off the
        # proof's latency (kernels on the auxiliary stream hidden behind a big launch have none)
        ev = []
        for n_, s_, e_ in seg:
            ev.append((s_, 1, short(n_)))
            ev.append((e_, -1, short(n_)))
        ev.sort(key=lambda x: (x[0], x[1]))
        active, excl, last = {}, {}, ev[0][0]
        for t_, d_, n_ in ev:
            if t_ > last:
                live = [k for k, c in active.items() if c > 0]
                if len(live) == 1 and active[live[0]] == 1:
                    excl[live[0]] = excl.get(live[0], 0) + (t_ - last)
                last = t_
            active[n_] = active.get(n_, 0) + d_
        # the FRI commit phase layer by layer: a layer = from one k_fri_fold to the next (its tree, the challenger step, the fold)
        folds = [(s_ - t0) / 1e3 for n_, s_, e_ in seg if "k_fri_fold" in n_]
        if len(folds) > 2:
            first_ro = min(((s_ - t0) / 1e3 for n_, s_, e_ in seg if "k_reduce_openings" in n_), default=folds[0])
            grind = min(((s_ - t0) / 1e3 for n_, s_, e_ in seg if "k_pow_grind" in n_), default=(t1 - t0) / 1e3)
            f.write("phases_us: up to the reduced openings %.0f, reduced openings .. first fold %.0f, FRI layers %.0f, proof of work .. end %.0f\n"
                    % (first_ro, folds[0] - first_ro, grind - folds[0], (t1 - t0) / 1e3 - grind))
            f.write("fri_layer_us (first fold -> next fold, ..., last fold -> proof of work): %s\n"
                    % " ".join("%.0f" % (b_ - a_) for a_, b_ in zip(folds, folds[1:] + [grind])))
        f.write("kernel launches total_us avg_us exclusive_us\n")
        for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            f.write("%-28s %5d %10.1f %8.1f %10.1f\n" % (k, n, t / 1e3, t / 1e3 / n, excl.get(k, 0) / 1e3))
    # the launch sequence itself (start and duration of every launch of that proof, and the queue it ran on where the view has one)
    try:
        cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
        qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
        q = "select name, start, end%s from kernels where start >= ? and start < ? order by start" % ((", " + qcol) if qcol else "")
        with open(out_txt.replace("_timeline_", "_launch_sequence_"), "w") as f:
            f.write("# %s\n# every launch of one complete proof: start_us duration_us %s kernel\n" % (header, qcol or "-"))
            for r in db.execute(q, (seg[0][1], rows[b][1])).fetchall():
                f.write("%9.1f %8.1f %s %s\n" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "-", short(r[0])))
    except Exception as e:  # noqa: BLE001 - an extra
        print("launch sequence failed:", e)


def counters(db_path):
    db = sqlite3.connect(db_path)
    out = {}
    for name, cname, n, s in db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        e = out.setdefault(short(name), {})
        a = e.get(cname, (0, 0.0))
        e[cname] = (a[0] + n, a[1] + s)  # template instances of one kernel (k_quotient<..>, k_col_dot<..>) are pooled
    dur = {}
    for name, n, s in db.execute("select kernel_name, count(distinct dispatch_id), 0 from counters_collection group by kernel_name"):
        dur[short(name)] = n
    return out, dur


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    # default: straight into profiles/ (tracked); on the GPU box pass an output directory under gpurun_out/ — the rocpd
    # databases themselves are too big to travel back, the summaries are not
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for leg, cmd in (("stats1", "python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 10 --warmup 1"),
                     ("stats2", "python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 3")):
        p = db_of(os.path.join(src, leg))
        if p:
            kernel_stats(p, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, "inflight1" if leg == "stats1" else "default")), "rocprofv3 --kernel-trace --stats -- " + cmd + "   (MI355X)")
    p = db_of(os.path.join(src, "stats1"))
    if p:
        try:
            timeline(p, os.path.join(dst, "%s_timeline_inflight1.txt" % tag), "rocprofv3 --kernel-trace -- python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 10 --warmup 1   (MI355X)")
        except sqlite3.Error as e:
            print("timeline failed:", e)
    for f in ("bench_default.json", "bench_inflight1.json", "bench_full.json", "bench_c3.json", "bench_c4.json", "bench_poseidon.json", "bench_c4_poseidon.json", "stats1.json", "stats2.json"):
        p = os.path.join(src, f)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, f)))
    pmc = {}
    p = db_of(os.path.join(src, "pmc_valu"))
    if p:
        c, _ = counters(p)
        for k, v in c.items():
            n = v["SQ_WAVES"][0]
            e = pmc.setdefault(k, {"launches_sampled": n})
            waves = v["SQ_WAVES"][1]
            e["waves_per_launch"] = waves / n
            e["valu_wave_instr_per_launch"] = v["SQ_INSTS_VALU"][1] / n
            e["valu_instr_per_wave"] = v["SQ_INSTS_VALU"][1] / waves if waves else 0
            e["salu_instr_per_wave"] = v["SQ_INSTS_SALU"][1] / waves if waves else 0
            wc = v["SQ_WAVE_CYCLES"][1]
            if wc:
                e["frac_wave_cycles_valu_active"] = v["SQ_ACTIVE_INST_VALU"][1] / wc
                e["frac_wave_cycles_wait_any"] = v["SQ_WAIT_ANY"][1] / wc
                e["frac_wave_cycles_wait_inst_any"] = v["SQ_WAIT_INST_ANY"][1] / wc
                e["frac_wave_cycles_active_inst_any"] = v["SQ_ACTIVE_INST_ANY"][1] / wc
    p = db_of(os.path.join(src, "pmc_busy"))
    if p:
        c, _ = counters(p)
        for k, v in c.items():
            try:
                n = v["GRBM_GUI_ACTIVE"][0]
                # Calibrated on tools/microbench.hip's rate kernels (profiles/r02_counter_calibration.txt, second table):
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs (value / 8 = GPU cycles of the launch); SQ_ACTIVE_INST_VALU equals
                # SQ_INSTS_VALU (one unit per instruction, whatever its issue rate) and SQ_BUSY_CU_CYCLES / 256 = cycles a CU was busy.
                cycles = v["GRBM_GUI_ACTIVE"][1] / n / 8.0
                insts = v["SQ_INSTS_VALU"][1] / n
                e = pmc.setdefault(k, {})
                e["gpu_cycles_per_launch"] = cycles
                # SIMD cycles available per VALU instruction issued: 2.3-2.5 = a kernel of full-rate instructions running at the issue
                # peak, 4.2 = one of half-rate instructions at ITS peak (profiles/r02_microbench.txt); larger = issue slots left idle
                e["simd_cycles_per_valu_instr"] = cycles * 1024.0 / insts if insts else None
                e["cu_busy_fraction"] = v["SQ_BUSY_CU_CYCLES"][1] / n / 256.0 / cycles if cycles else None
            except (KeyError, ZeroDivisionError):
                pass
    for leg, cname, key in (("pmc_fetch", "FETCH_SIZE", "hbm_read_bytes_per_launch"), ("pmc_write", "WRITE_SIZE", "hbm_write_bytes_per_launch")):
        p = db_of(os.path.join(src, leg))
        if p:
            c, _ = counters(p)
            for k, v in c.items():
                n, s = v[cname]
                # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B
                # for wide coalesced reads (MI355X_MICROARCH.md, HBM section): doubled here, raw value kept beside it
                raw = s / n * 1024.0
                e = pmc.setdefault(k, {})
                e[key + "_raw"] = raw
                e[key] = raw * (2.0 if cname == "FETCH_SIZE" else 1.0)
    if pmc:
        with open(os.path.join(dst, tag + "_pmc.json"), "w") as f:
            json.dump({"source": "tools/profile_round.sh passes pmc_valu / pmc_fetch / pmc_write (one proof in flight, 1 warmup + 2 timed proofs + setup)",
                       "notes": "SQ_* cycle counters are in quad-cycles; FETCH_SIZE doubled per the gfx950 correction (calibrated this round on copies of known size, "
                                "profiles/r02_counter_calibration.txt: x2 holds for 4- and 16-byte-per-lane coalesced reads and for 128-byte row segments; 64-byte row "
                                "segments really fetch two bytes per byte used; WRITE_SIZE is exact), *_raw = as reported",
                       "kernels": pmc}, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
