#!/usr/bin/env python3
"""Per-(kernel, grid size) statistics of a rocprofv3 kernel trace (rocpd sqlite): small launches of one kernel differ
only in their grid, so grouping by it tells the GEMM shapes of a decoder layer apart.
    python profiles/by_grid.py <trace.db> [name substring]"""
import re
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size")][0]
    wx = [c for c in cols if c.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size")][0]
    by = {}
    for name, g, w, d in cur.execute(f"select name, {gx}, {wx}, end - start from kernels"):
        if pat not in name:
            continue
        m = re.search(r"(\w+(<[^(]*>)?)\(", name.replace("(anonymous namespace)::", "").replace("ltr::", ""))
        by.setdefault((m.group(1) if m else name[:60], int(g) // max(int(w), 1)), []).append(d)
    tot = sum(sum(v) for v in by.values()) or 1
    for (k, g), d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        d.sort()
        print(f"{100.0 * sum(d) / tot:5.1f}%  {k[:58]:58s} wgs {g:6d}  n {len(d):4d}  median {d[len(d) // 2] / 1e3:7.1f} us  "
              f"min {d[0] / 1e3:7.1f}  max {d[-1] / 1e3:7.1f}")


if __name__ == "__main__":
    main()
