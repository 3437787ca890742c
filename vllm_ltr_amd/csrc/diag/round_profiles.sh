#!/bin/bash
# Everything under profiles/ that describes the shipped kernels of a round, in one GPU call:
#   bash vllm_ltr_amd/csrc/diag/round_profiles.sh r04_v1     -> gpurun_out/r04_v1/*  (copy what is wanted to profiles/)
TAG=${1:-r04_v1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
bash $R/vllm_ltr_amd/csrc/diag/refresh_profiles.sh $TAG > $O/refresh.log 2>&1
cd $R
python bench.py --model 350m --profile lmsys --steps 3 --warmup 1 --no-strong --no-scale-points --no-class-head > $O/bench_config3.json 2> $O/bench_config3.err
# the reference's own fp16 arithmetic as a second, separately labelled number (LTR_F_ONE_PASS; never the headline)
python bench.py --weight-dtype f16-1pass --no-cpu-baseline --no-class-head > $O/bench_1pass.json 2> $O/bench_1pass.err
# driver / workers mode of the sharded call, dry run of the protocol with two ranks on the one device (gloo control plane)
LTR_BENCH_ONE_DEVICE=1 LTR_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 --queue 4096 --driver-broadcast --no-cpu-baseline --no-unfused --no-strong --no-scale-points --no-class-head --steady-new 0 > $O/bench_driver_broadcast_dryrun.json 2> $O/bench_driver_broadcast_dryrun.err
LTR_BENCH_ONE_DEVICE=1 LTR_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 --queue 4096 --no-cpu-baseline --no-unfused --no-strong --no-scale-points --no-class-head --steady-new 0 > $O/bench_spmd_dryrun.json 2> $O/bench_spmd_dryrun.err
python bench.py --scale-table --steps 2 --warmup 1 --no-cpu-baseline --no-unfused --no-strong --no-scale-points --no-class-head --steady-new 0 > $O/scale_table.jsonl 2> $O/scale_table.err
python bench.py --trace burst > $O/trace_burst.json 2>/dev/null
python bench.py --trace gamma > $O/trace_gamma.json 2>/dev/null
python bench.py --trace gamma --trace-rate 64 > $O/trace_gamma64.json 2>/dev/null
python bench.py --trace burst --model 350m > $O/trace_burst_350m.json 2>/dev/null
python bench.py --trace gamma --model 350m > $O/trace_gamma_350m.json 2>/dev/null
python bench.py --train --steps 10 --warmup 3 > $O/train_bench.json 2>/dev/null
python tests/diag/small_call_profile.py 1 2 4 8 16 32 64 128 256 > $O/small_calls.txt 2>&1
ls -la $O | head -40
