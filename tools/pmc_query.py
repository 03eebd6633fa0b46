#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc run (rocpd SQLite) per kernel:  python tools/pmc_query.py <dir-or-db>
Prints, for every kernel, the sum of each collected counter (template instances pooled).  Run it on the GPU box and
keep the text: the databases themselves can exceed what gpurun copies back."""
import glob
import json
import os
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    out = {}
    for p in dbs:
        db = sqlite3.connect(p)
        for name, c, n, s in db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
            m = re.search(r"(k_[a-z0-9_]+)", name)
            k = m.group(1) if m else name.split("(")[0][:40]
            e = out.setdefault(k, {})
            e[c] = e.get(c, 0.0) + s
            e["_dispatches"] = max(e.get("_dispatches", 0), n)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
