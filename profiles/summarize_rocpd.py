#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a per-kernel stats
CSV: name, calls, total ns, avg ns, min, max, percent.  Usage:
    python profiles/summarize_rocpd.py gpurun_out/prof_x/<host>/<pid>_results.db out.csv
(The same numbers `rocprofv3 --kernel-trace --stats --output-format csv` writes to
*_kernel_stats.csv.)"""
import csv
import sqlite3
import sys


def main(db, out):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), "
                       f"max(end - start) from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), f"{r[3]:.1f}", int(r[4]), int(r[5]), f"{100.0 * r[2] / total:.2f}"])
    for r in rows[:14]:
        print(f"{100.0 * r[2] / total:6.2f}%  calls={r[1]:6d}  avg={r[3] / 1e3:9.1f}us  {r[0][:90]}")
    # template instances of one kernel taken together (gemm_f16s_kernel<0|1|2>: plain / LayerNorm producer / consumer epilogue)
    import re
    comb = {}
    for r in rows:
        m = re.search(r"(\w+)<[^(]*>\(", r[0])
        if m:
            c = comb.setdefault(m.group(1), [0, 0])
            c[0] += r[1]; c[1] += r[2]
    for k, (n, t) in comb.items():
        print(f"combined {k}<*>: calls={n} avg={t / n / 1e3:.1f}us total={t / 1e6:.2f}ms")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
