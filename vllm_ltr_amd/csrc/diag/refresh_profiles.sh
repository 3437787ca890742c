# Regenerates (on the GPU box, into gpurun_out/) the artefacts kept under profiles/: the PMC traffic of the GEMM
# (first, because bench.py reads profiles/gemm_traffic.json for roofline.traffic), the default bench line and the
# rocprofv3 kernel trace summary of the same command.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_w.log 2>&1
cd $R
python profiles/make_gemm_traffic.py $(find gpurun_out/pmc_fetch -name "*.db" | head -1) $(find gpurun_out/pmc_write -name "*.db" | head -1)
cp profiles/gemm_traffic.json gpurun_out/gemm_traffic_v4.json
python bench.py > gpurun_out/bench_v4.json 2> gpurun_out/bench_v4.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v4 -o v4 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_v4.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(find gpurun_out/prof_v4 -name "*.db" | head -1) gpurun_out/v4_kernel_stats.csv
find gpurun_out -name "*.db" -delete
tail -2 gpurun_out/bench_v4.json
