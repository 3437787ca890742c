#!/bin/bash
# Regenerates (on the GPU box, into gpurun_out/<tag>/) every artefact kept under profiles/ for the shipped kernels:
#   1. PMC calibration of FETCH_SIZE / WRITE_SIZE on known-byte kernels (diag/pmc_calib.hip)
#   2. PMC traffic of every kernel of the bench call  -> kernel_traffic.json (+ gemm_traffic.json, read by bench.py)
#   3. SQ counters (MFMA busy, LDS bank conflicts, wave cycles) -> pmc_sq.txt
#   4. the default bench line                          -> bench.json
#   5. rocprofv3 --kernel-trace --stats of the same command -> kernel_stats.csv
# usage: bash vllm_ltr_amd/csrc/diag/refresh_profiles.sh <tag> [bench workload flags]     (tag e.g. r02_v8); copy
#        gpurun_out/<tag>/* to profiles/.  BASELINE config 3: ... <tag> --model 350m --profile lmsys with
#        LTR_PROFILE_WORKLOAD=350m/lmsys/8192 (the PMC artefacts then go to profiles/gemm_traffic.350m_lmsys_8192.json etc.)
TAG=${1:-r02}
shift
WL="$@"
SUF=$(python -c "import os; w=os.environ.get('LTR_PROFILE_WORKLOAD','125m/sharegpt/8192'); print('' if w=='125m/sharegpt/8192' else '.'+w.replace('/','_'))")
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
export LTR_PROFILE_TAG=$TAG          # stamped into gemm_traffic.json / gemm_pmc.json with the hash of the kernel sources
set -x
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/vllm_ltr_amd/csrc/diag/pmc_calib.hip -o /tmp/pmc_calib || exit 1
rocprofv3 --pmc FETCH_SIZE -d $O/calib_fetch -o f -- /tmp/pmc_calib > $O/calib_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/calib_write -o w -- /tmp/pmc_calib > $O/calib_w.log 2>&1
BENCH="python $R/bench.py $WL --steps 1 --warmup 0 --no-cpu-baseline --no-unfused --no-profile-pass --no-strong --no-scale-points --no-class-head --no-config3 --no-host-inclusive --steady-new 0"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- $BENCH > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- $BENCH > $O/pmc_w.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o s -- $BENCH > $O/pmc_s.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_l2 -o l -- $BENCH > $O/pmc_l.log 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python profiles/make_traffic.py $(db pmc_fetch) $(db pmc_write) $(db calib_fetch) $(db calib_write) $O/kernel_traffic.json > $O/traffic.log 2>&1
cp profiles/gemm_traffic$SUF.json $O/gemm_traffic$SUF.json
python profiles/summarize_pmc.py $(db pmc_sq) $(db pmc_l2) > $O/pmc_sq.txt 2>&1
python profiles/make_gemm_pmc.py $(db pmc_sq) $(db pmc_l2) $O/gemm_pmc$SUF.json > /dev/null 2>&1
cp $O/gemm_pmc$SUF.json profiles/gemm_pmc$SUF.json
python bench.py $WL > $O/bench.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py $WL --steps 2 --warmup 1 --no-cpu-baseline --no-unfused --no-profile-pass --no-strong --no-scale-points --no-class-head --no-config3 --no-host-inclusive --steady-new 0 > $O/prof.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(db prof) $O/kernel_stats.csv > $O/kernel_stats.txt 2>&1
find $O -name "*.db" -delete
tail -1 $O/bench.json | cut -c1-600
cat $O/kernel_stats.txt
