O=gpurun_out/r03_n; mkdir -p $O
LTR_GEMM_DEEP_TILES=600 LTR_GEMM_DEEP_M=1024 python -m pytest tests/test_gpu_small_batches.py tests/test_gpu_gemm_epilogue.py -m gpu -q -k "not latency" > $O/t1.log 2>&1; tail -3 $O/t1.log
for cfg in "0 0" "1024 320" "1024 600" "1024 1200" "3072 600" "3072 1200" "512 600"; do set -- $cfg; echo "DEEP_M=$1 DEEP_TILES=$2"; LTR_GEMM_DEEP_M=$1 LTR_GEMM_DEEP_TILES=$2 python tests/diag/small_call_profile.py 4 8 16 32 64 128 256 2>/dev/null | cut -c1-100; done | tee $O/ab_deep.txt
