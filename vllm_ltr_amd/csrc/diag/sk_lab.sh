# stream-K lab: (1) gemm_check with the stream-K kernel forced (several grid sizes -> 2-piece and many-piece tiles, multi-segment
# workgroups), (2) gemm_bench per row count: LTR_GEMM_SK=0 (round-4 choice) vs 3 (stream-K everywhere) vs 1 (the model's choice)
cd "$(dirname "$0")/.."; export LD_LIBRARY_PATH=$PWD:$LD_LIBRARY_PATH
C=build/gemm_check; B=build/gemm_bench
if [ "${SK_CHECK:-1}" = 1 ]; then
for wgs in 512 16 100; do for shape in "700 256 1024" "3500 256 1024" "1000 768 768" "2500 256 256" "129 1024 4096"; do
  echo "== check SK=3 WGS=$wgs $shape"; LTR_GEMM_SK=3 LTR_GEMM_SK_WGS=$wgs timeout 300 $C $shape | grep -v "^a'\|^out\[\|^stats\[\|^lnc\[" | tr '\n' ' '; echo
done; done
echo "== check window SK=3: 2000 256 1024 rows 1536+464"; LTR_GEMM_SK=3 timeout 300 $C 2000 256 1024 1536 464 | tr '\n' ' '; echo
fi
for HF in "768 3072" ${SK_350:+"1024 4096"}; do for M in ${MS:-874 1382 2171 3947 5928 12765 23078 50000}; do for sk in 0 3 1; do
  echo "== bench M=$M HF=$HF SK=$sk"; LTR_GEMM_SK=$sk timeout 120 $B $M $HF 40 | grep -v "^layer"
done; done; done
