#!/bin/bash
# Per-workgroup phase stamps of the small-batch GEMM kernel: builds a private copy of the library with -DLTR_GEMM_TIMELINE and
# runs diag/small_timeline.hip for the four shapes of a decoder layer at M rows.   usage: run_small_timeline.sh [M] [H] [F]
set -e
cd "$(dirname "$0")/.."
M=${1:-262}; H=${2:-768}; F=${3:-3072}
T=${TMPDIR:-/tmp}/ltr_stl; mkdir -p $T
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DLTR_GEMM_TIMELINE -DLTR_ATTN_TIMELINE"
for s in ltr_api ltr_rank ltr_rows ltr_gemm ltr_attn ltr_pool ltr_head ltr_train ltr_trainer; do /opt/rocm/bin/hipcc $FL -c $s.hip -o $T/$s.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $T/libltr_tl.so $T/*.o
/opt/rocm/bin/hipcc $FL diag/small_timeline.hip -L$T -lltr_tl -Wl,-rpath,$T -o $T/small_timeline
/opt/rocm/bin/hipcc $FL diag/attn_small_timeline.hip -L$T -lltr_tl -Wl,-rpath,$T -o $T/attn_small_timeline
for L in 64 262 1024; do $T/attn_small_timeline $L; done
for mm in $M; do
$T/small_timeline $mm $((3*H)) $H
$T/small_timeline $mm $H $H
$T/small_timeline $mm $F $H
$T/small_timeline $mm $H $F
done
