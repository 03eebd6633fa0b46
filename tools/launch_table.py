#!/usr/bin/env python3
"""Per-launch view of a rocprofv3 --kernel-trace database (rocpd SQLite): for every kernel, launches grouped by grid size
with their average / total duration — separates the few big launches of a proof from its many tiny ones.
    python tools/launch_table.py <dir with the .db> [kernel-substring ...]"""
import glob
import os
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0]


def main():
    hits = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)
    db = sqlite3.connect(hits[0])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    print("# columns of view `kernels`:", cols)
    gcols = [c for c in ("grid_x", "grid_y", "grid_z", "workgroup_x", "lds_size", "scratch_size", "vgpr_count", "accum_vgpr_count", "sgpr_count") if c in cols]
    sel = ", ".join(["name", "start", "end"] + gcols)
    rows = db.execute("select %s from kernels" % sel).fetchall()
    want = sys.argv[2:]
    table = {}
    for r in rows:
        k = short(r[0])
        if want and not any(w in k for w in want):
            continue
        key = (k,) + tuple(r[3:])
        t = table.setdefault(key, [0, 0.0])
        t[0] += 1
        t[1] += (r[2] - r[1]) / 1e3
    print("kernel", *gcols, "launches", "avg_us", "total_us")
    for key, (n, tot) in sorted(table.items(), key=lambda kv: -kv[1][1])[:80]:
        print(*key, n, "%.1f" % (tot / n), "%.0f" % tot)


if __name__ == "__main__":
    main()
