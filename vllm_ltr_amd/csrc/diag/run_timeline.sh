#!/bin/bash
# Per-workgroup timeline of the production GEMM: builds a private copy of the library with
# -DLTR_GEMM_TIMELINE (plus any extra flags in $1) and runs diag/gemm_timeline.hip against it.
set -e
cd "$(dirname "$0")/.."
T=${TMPDIR:-/tmp}/ltr_tl; mkdir -p $T
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DLTR_GEMM_TIMELINE $1"
for s in ltr_api ltr_rank ltr_rows ltr_gemm ltr_attn ltr_pool ltr_head ltr_train; do /opt/rocm/bin/hipcc $F -c $s.hip -o $T/$s.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $T/libltr_tl.so $T/*.o
/opt/rocm/bin/hipcc $F diag/gemm_timeline.hip -L$T -lltr_tl -Wl,-rpath,$T -o $T/gemm_timeline
$T/gemm_timeline
