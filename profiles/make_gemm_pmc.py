#!/usr/bin/env python3
"""MFMA utilisation, L2 hit rate and effective clock of the dominant kernels from the SQ / GRBM / TCC passes of
diag/refresh_profiles.sh (rocprofv3 rocpd sqlite):

    python profiles/make_gemm_pmc.py <sq.db> <grbm_tcc.db> [out.json]

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles), cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the
8 XCDs); 16 busy cycles per v_mfma_f32_16x16x32_f16.  Profiled passes clock ~3 % lower than plain runs (MI355X_MICROARCH.md,
DVFS give-back), so the fraction is of the profiled launch."""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_traffic import source_stamp  # noqa: E402


def table(db):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, c, v, d in cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        m = re.search(r"(\w+)(<[^(]*>)?\(", name.replace("(anonymous namespace)::", "").replace("ltr::", ""))
        k = m.group(1) if m else name
        e = out.setdefault(k, {}).setdefault(c, [0.0, set()])
        e[0] += v
        e[1].add(d)
    return {k: {c: e[0] / len(e[1]) for c, e in cs.items()} for k, cs in out.items()}


def main():
    sq, l2 = table(sys.argv[1]), table(sys.argv[2])
    out_path = sys.argv[3] if len(sys.argv) > 3 else "profiles/gemm_pmc.json"
    res = {}
    for k in ("gemm_f16s_kernel", "attn_f16s_kernel"):
        if k not in sq or k not in l2:
            continue
        cyc = l2[k]["GRBM_GUI_ACTIVE"] / 8.0
        res[k] = dict(cycles_per_launch=cyc,
                      mfma_busy=sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc),
                      lds_bank_conflict_frac=sq[k]["SQ_LDS_BANK_CONFLICT"] / max(sq[k]["SQ_LDS_IDX_ACTIVE"], 1.0),
                      wave_parked_frac=sq[k]["SQ_WAIT_ANY"] / sq[k]["SQ_WAVE_CYCLES"],
                      wave_issue_stall_frac=sq[k]["SQ_WAIT_INST_ANY"] / sq[k]["SQ_WAVE_CYCLES"],
                      l2_hit_rate=l2[k]["TCC_HIT_sum"] / (l2[k]["TCC_HIT_sum"] + l2[k]["TCC_MISS_sum"]))
    res["source"] = source_stamp()
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
