#!/usr/bin/env python3
"""The last scoring call of a rocprofv3 kernel trace (rocpd sqlite), kernel by kernel on the GPU's clock: start -> end (duration),
gap to the previous kernel's end, workgroups, kernel.  A call starts at its embed_gather_kernel.
    python profiles/k1_timeline.py <trace.db> [which call from the end, default 2 = the last one issued in the loop]"""
import re
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size")][0]
    wx = [c for c in cols if c.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size")][0]
    gy = [c for c in cols if c.lower() in ("grid_y", "grid_size_y")]
    wy = [c for c in cols if c.lower() in ("workgroup_y", "workgroup_size_y")]
    extra = f", {gy[0]}, {wy[0]}" if gy and wy else ", 1, 1"
    rows = list(cur.execute(f"select name, {gx}, {wx}, start, end{extra} from kernels order by start"))
    starts = [i for i, r in enumerate(rows) if "embed_gather_kernel" in r[0]]
    if len(starts) < back:
        print("not enough calls in the trace"); return
    i0 = starts[-back]
    i1 = starts[-back + 1] if back > 1 else len(rows)
    call = rows[i0:i1]
    t0 = call[0][3]
    prev_end = t0
    gaps = 0.0
    for name, g, w, s, e, g2, w2 in call:
        m = re.search(r"(\w+(<[^(]*>)?)\(", name.replace("(anonymous namespace)::", "").replace("ltr::", ""))
        short = (m.group(1) if m else name)[:70]
        gap = (s - prev_end) / 1e3
        gaps += max(gap, 0.0)
        wgs = (int(g) // max(int(w), 1)) * (int(g2) // max(int(w2), 1))
        print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} ({(e - s) / 1e3:5.1f})  gap {gap:5.1f}  wgs {wgs:5d}  {short}")
        prev_end = e
    print(f"# {len(call)} kernels, span {(call[-1][4] - t0) / 1e3:.1f} us, kernel time {sum(r[4] - r[3] for r in call) / 1e3:.1f} us, gaps {gaps:.1f} us")


if __name__ == "__main__":
    main()
