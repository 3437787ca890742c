#!/usr/bin/env python3
"""Per-kernel PMC counter sums/averages from a rocprofv3 rocpd sqlite db (ROCm 7.2).
    python profiles/summarize_pmc.py <db> [<db> ...]"""
import sqlite3
import sys
from collections import defaultdict


def main(paths):
    for db in paths:
        con = sqlite3.connect(db)
        cur = con.cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        name = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "kernel" in c][0]
        cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
        val = "value" if "value" in cols else "counter_value"
        disp = "dispatch_id" if "dispatch_id" in cols else cols[0]
        agg = defaultdict(lambda: defaultdict(float))
        nd = defaultdict(set)
        for k, c, v, d in cur.execute(f"select {name}, {cname}, {val}, {disp} from counters_collection"):
            import re
            m = re.search(r"(\w+)(<[^(]*>)?\(", k.replace("(anonymous namespace)", "anon"))
            short = (m.group(1) if m else k)[:40]
            agg[short][c] += v
            nd[short].add(d)
        for k in agg:
            print(f"== {k}  dispatches={len(nd[k])}")
            for c, v in sorted(agg[k].items()):
                print(f"   {c:32s} total={v:16.0f}  per-dispatch={v / max(len(nd[k]), 1):14.1f}")


if __name__ == "__main__":
    main(sys.argv[1:])
