#!/usr/bin/env python3
"""HBM-side traffic per kernel launch from rocprofv3 PMC passes, with the FETCH_SIZE / WRITE_SIZE
calibration measured on known-byte kernels in the library's own access forms.

    python profiles/make_traffic.py <bench_fetch.db> <bench_write.db> <calib_fetch.db> <calib_write.db> [out.json]

* bench_*.db : `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` over `python bench.py --steps 1 --warmup 0`
  (the two counters cannot share a pass: 3 + 2 of the 4 TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").
* calib_*.db : the same two passes over diag/pmc_calib (4 GiB touched exactly once per kernel).

Both counters are in KiB.  factor[form] = known bytes / reported bytes; the guide's gfx950 rule (wide
coalesced reads report 1/2) is what the dma / reg_x4 factors should reproduce.  Writes
profiles/kernel_traffic.json (and keeps profiles/gemm_traffic.json, which bench.py reads, in sync)."""
import hashlib
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def workload_suffix():
    """'' for the headline workload (125m / sharegpt / 8192 per GPU), '.<model>_<profile>_<n>' otherwise: the PMC artefacts
    of another workload (BASELINE config 3: 350m / lmsys / 8192) live in their own files (bench.py picks by workload)."""
    w = os.environ.get("LTR_PROFILE_WORKLOAD", "125m/sharegpt/8192")
    return "" if w == "125m/sharegpt/8192" else "." + w.replace("/", "_")


def source_stamp():
    """Which kernels the pass measured: LTR_PROFILE_TAG (refresh_profiles.sh's tag) + the hash of the kernel sources
    (bench.py uses the numbers only while the hash still matches - same function there, kernel_sources_sha16)."""
    h = hashlib.sha256()
    for rel in ("vllm_ltr_amd/csrc/ltr_gemm.hip", "vllm_ltr_amd/csrc/ltr_api.hip"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    # the passes of diag/refresh_profiles.sh run the default bench workload: bench.py quotes them for that workload only
    return dict(tag=os.environ.get("LTR_PROFILE_TAG", ""), kernel_sha16=h.hexdigest()[:16],
                workload=os.environ.get("LTR_PROFILE_WORKLOAD", "125m/sharegpt/8192"))

CALIB_BYTES = 4 << 30
# which calibrated access form dominates each production kernel's reads
READ_FORM = {"gemm_f16s_kernel": "calib_dma_contig", "attn_f16s_kernel": "calib_dma_rows128",
             "layernorm_kernel": "calib_reg_x4", "embed_gather_kernel": "calib_reg_x2",
             "pool_head_kernel": "calib_reg_x4", "gather_last_rows_kernel": "calib_reg_x4"}
WRITE_FORM = {"attn_f16s_kernel": "calib_store_x2"}


def short(name):
    m = re.search(r"(\w+)(<[^(]*>)?\(", name.replace("(anonymous namespace)::", "").replace("ltr::", ""))
    return m.group(1) if m else name


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    tot, disp = {}, {}
    for name, v, d in cur.execute("select kernel_name, value, dispatch_id from counters_collection where counter_name=?",
                                  (counter,)):
        k = short(name)
        tot[k] = tot.get(k, 0.0) + v
        disp.setdefault(k, set()).add(d)
    return {k: (tot[k] / len(disp[k]), len(disp[k])) for k in tot}


def main():
    bf, bw, cf, cw = sys.argv[1:5]
    out_path = sys.argv[5] if len(sys.argv) > 5 else "profiles/kernel_traffic.json"
    calib_f, calib_w = per_kernel(cf, "FETCH_SIZE"), per_kernel(cw, "WRITE_SIZE")
    factors = {}
    for k, (kib, _) in calib_f.items():
        if k.startswith("calib_") and "store" not in k and kib > 0:
            factors[k] = CALIB_BYTES / (kib * 1024.0)
    for k, (kib, _) in calib_w.items():
        if k.startswith("calib_store") and kib > 0:
            factors[k] = CALIB_BYTES / (kib * 1024.0)
    fetch, write = per_kernel(bf, "FETCH_SIZE"), per_kernel(bw, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f_kib, nf = fetch.get(k, (0.0, 0))
        w_kib, nw = write.get(k, (0.0, 0))
        rf = factors.get(READ_FORM.get(k, "calib_reg_x4"), 2.0)
        wf = factors.get(WRITE_FORM.get(k, "calib_store_x4"), 1.0)
        kernels[k] = dict(launches_profiled=[nf, nw], FETCH_SIZE_KiB_per_launch_raw=f_kib,
                          WRITE_SIZE_KiB_per_launch_raw=w_kib, read_factor=rf, write_factor=wf,
                          read_bytes_per_launch=f_kib * 1024.0 * rf, write_bytes_per_launch=w_kib * 1024.0 * wf,
                          hbm_bytes_per_launch=f_kib * 1024.0 * rf + w_kib * 1024.0 * wf)
    out = dict(source=source_stamp(),
               calibration=dict(bytes_per_kernel=CALIB_BYTES, factor=factors,
                                note="factor = known bytes / counter bytes on diag/pmc_calib.hip (4 GiB touched once "
                                     "per kernel, > Infinity Cache); fabric-side counters: Infinity-Cache hits are "
                                     "included, so production numbers are an upper bound on DRAM bytes"),
               kernels=kernels)
    json.dump(out, open(out_path, "w"), indent=1)
    g = kernels.get("gemm_f16s_kernel")
    if g:
        json.dump(dict(kernel="gemm_f16s", launches_profiled=g["launches_profiled"],
                       FETCH_SIZE_KiB_per_launch_raw=g["FETCH_SIZE_KiB_per_launch_raw"],
                       WRITE_SIZE_KiB_per_launch=g["WRITE_SIZE_KiB_per_launch_raw"],
                       fetch_correction=g["read_factor"], write_correction=g["write_factor"],
                       hbm_bytes_per_launch=g["hbm_bytes_per_launch"], source=source_stamp(),
                       note="see kernel_traffic.json; corrections measured by diag/pmc_calib.hip"),
                  open(f"profiles/gemm_traffic{workload_suffix()}.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
