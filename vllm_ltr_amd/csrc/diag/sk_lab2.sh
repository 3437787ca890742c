cd "$(dirname "$0")/.."; R=$PWD
C=build/gemm_check; B=build/gemm_bench
export LD_LIBRARY_PATH=$R
for wgs in 512 16 100; do for shape in "700 256 1024" "3500 256 1024" "1000 768 768" "129 1024 4096"; do
  echo "== check SK=3 WGS=$wgs $shape"; LTR_GEMM_SK=3 LTR_GEMM_SK_WGS=$wgs timeout 300 $C $shape | grep -v "^a'\|^out\[\|^stats\[\|^lnc\[" | tr '\n' ' '; echo
done; done
echo "== check window SK=3: 2000 256 1024 rows 1536+464"; LTR_GEMM_SK=3 timeout 300 $C 2000 256 1024 1536 464 | grep -v "^a'\|^out\[\|^stats\[\|^lnc\[" | tr '\n' ' '; echo
for M in 2171 5928 23078; do
  for v in "" var1; do
    export LD_LIBRARY_PATH=$R/build/$v:$R
    for wgs in 512 256; do
    echo "== bench M=$M SK=3 variant=[$v] wgs=$wgs"; LTR_GEMM_SK=3 LTR_GEMM_SK_WGS=$wgs timeout 120 $B $M 768 3072 40 | grep -v "^layer"
    done
  done
done
