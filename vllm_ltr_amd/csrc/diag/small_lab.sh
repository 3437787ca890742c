#!/bin/bash
# Small-batch GEMM lab (round 4): per-shape time (launches back to back in the stream) of the four dense layers of a decoder
# layer at the row counts of a scheduler step with k = 1 ... 256 arrivals, under the launch_gemm A/B knobs: tile config
# (LTR_GEMM_FORCE_CFG: -1 = 128x256 two-stage kernel; small-batch kernel 0 = 32x64, 1 = 64x128, 2 = 128x256 behind a ring of
# four, 3 = 32x128, 4 = 64x64, 5 = 64x256), split-K parts of the narrow outputs (LTR_GEMM_FORCE_SPLIT) and the XCD map
# (LTR_GEMM_SMALL_MAP).
#   bash vllm_ltr_amd/csrc/diag/small_lab.sh > gpurun_out/lab.txt      (from the repository root, on the GPU box)
B=vllm_ltr_amd/csrc/build/gemm_bench
H=${H:-768}; F=${F:-3072}
CFGS=${CFGS:-"0 1 3 4 5 -1"}; SPS=${SPS:-"1 2 4"}; MAPS=${MAPS:-"3"}
for M in ${MS:-262 874 2171 5928 23078}; do
  echo "===== M=$M H=$H F=$F"
  run() { echo "--- $*"; env "$@" $B $M $H $F 40 | grep -v "^layer" ; }
  for cfg in $CFGS; do
    for map in $MAPS; do
      for sp in $SPS; do
        run LTR_GEMM_FORCE_CFG=$cfg LTR_GEMM_SMALL_MAP=$map LTR_GEMM_FORCE_SPLIT=$sp
      done
    done
  done
done
