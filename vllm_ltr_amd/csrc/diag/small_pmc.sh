#!/bin/bash
# PMC passes over the small-batch GEMMs (one shape per process: counters are summed per kernel name): L2 hit rate,
# fabric-side read bytes (FETCH_SIZE: x2 for wide reads, profiles/README.md), busy cycles.   usage: small_pmc.sh <outdir>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=${1:-$R/gpurun_out/small_pmc}
case $O in /*) ;; *) O=$R/$O ;; esac
mkdir -p $O
export TMPDIR=/tmp
B=$R/vllm_ltr_amd/csrc/build/gemm_bench
cd /tmp
for M in ${MS:-262 2171 5928}; do
  for sh in qkv out_proj fc1 fc2; do
    for pass in l2 fetch; do
      if [ $pass = l2 ]; then C="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; else C="FETCH_SIZE"; fi
      BENCH_ONLY=$sh rocprofv3 --pmc $C -d $O/${M}_${sh}_$pass -o p -- $B $M ${H:-768} ${F:-3072} 10 > $O/${M}_${sh}_$pass.log 2>&1
    done
    BENCH_ONLY=$sh rocprofv3 --kernel-trace --stats -d $O/${M}_${sh}_kt -o k -- $B $M ${H:-768} ${F:-3072} 10 > $O/${M}_${sh}_kt.log 2>&1
  done
done
cd $R
for M in ${MS:-262 2171 5928}; do
  for sh in qkv out_proj fc1 fc2; do
    echo "##### M=$M $sh"
    python profiles/summarize_pmc.py $(find $O/${M}_${sh}_l2 -name "*.db" | head -1) $(find $O/${M}_${sh}_fetch -name "*.db" | head -1) 2>&1 | grep -v "pack_weight\|fill_\|relu_planes" 
    python profiles/summarize_rocpd.py $(find $O/${M}_${sh}_kt -name "*.db" | head -1) /dev/null 2>&1 | grep -i "gemm\|splitk" | head -4
  done
done > $O/summary.txt 2>&1
find $O -name "*.db" -delete
