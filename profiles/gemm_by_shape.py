#!/usr/bin/env python3
"""Per-shape statistics of the GEMM launches in a rocprofv3 kernel trace (rocpd sqlite): the four dense layers of
a decoder layer differ in grid size, so grouping by grid tells them apart.
    python profiles/gemm_by_shape.py <trace.db> [tokens_per_pass=196608] [H=768] [F=3072]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 196608
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 768
    F = int(sys.argv[4]) if len(sys.argv) > 4 else 3072
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size")][0]
    rows = cur.execute(f"select name, {gx}, end - start from kernels where name like '%gemm_f16s%'").fetchall()
    by = {}
    for _, g, d in rows:
        by.setdefault(int(g), []).append(d)
    tiles_m = (T + 127) // 128
    shapes = {tiles_m * (3 * H // 256) * 512: ("qkv", 3 * H, H), tiles_m * (H // 256) * 512: ("out/fc2", H, None),
              tiles_m * (F // 256) * 512: ("fc1", F, H)}
    for g in sorted(by, key=lambda k: -sum(by[k])):
        d = sorted(by[g])
        print(f"grid {g:9d} ({g // 512:6d} tiles)  launches {len(d):4d}  median {d[len(d)//2]/1e3:8.1f} us  "
              f"min {d[0]/1e3:8.1f}  max {d[-1]/1e3:8.1f}  total {sum(d)/1e6:8.2f} ms  {shapes.get(g, ('?',))[0]}")


if __name__ == "__main__":
    main()
