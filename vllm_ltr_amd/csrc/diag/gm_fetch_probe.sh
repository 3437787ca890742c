#!/bin/bash
# Cold-call probe (round 4): fabric-side read bytes (FETCH_SIZE x 2) and time of the wide K = 768 GEMMs of a
# full-size pass under different tile-group orders (LTR_GEMM_GM: group size, + 65536 = N fastest inside the group).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-gm_probe}; mkdir -p $O; export TMPDIR=/tmp
B=$R/vllm_ltr_amd/csrc/build/gemm_bench
cd /tmp
for gm in 8 4 16 2 65544 65540 65538 65552; do
  for sh in qkv fc1; do
    LTR_GEMM_GM=$gm BENCH_ONLY=$sh rocprofv3 --pmc FETCH_SIZE -d $O/${gm}_${sh}_f -o p -- $B 196608 768 3072 6 > /dev/null 2>&1
    LTR_GEMM_GM=$gm BENCH_ONLY=$sh rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/${gm}_${sh}_l -o p -- $B 196608 768 3072 6 > /dev/null 2>&1
    f=$(python $R/profiles/summarize_pmc.py $(find $O/${gm}_${sh}_f -name "*.db" | head -1) | grep -A1 "gemm_f16s_kernel" | grep FETCH | awk '{print $3}' | sed 's/per-dispatch=//')
    l=$(python $R/profiles/summarize_pmc.py $(find $O/${gm}_${sh}_l -name "*.db" | head -1) | grep -A2 "gemm_f16s_kernel" | grep "TCC" | awk '{print $2, $3}' | tr '\n' ' ')
    echo "GM=$gm $sh FETCH $f | $l"
  done
  echo "GM=$gm time:"; LTR_GEMM_GM=$gm $B 196608 768 3072 10 | grep "qkv\|fc1"
done
find $O -name "*.db" -delete
