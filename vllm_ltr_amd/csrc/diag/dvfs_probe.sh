B=vllm_ltr_amd/csrc/build/gemm_bench
for reps in 20 200 2000; do for M in 262 5928; do echo "== M=$M reps=$reps"; $B $M 768 3072 $reps | grep -v "^layer"; done; done
rocm-smi --showclocks 2>/dev/null | head -20
( $B 5928 768 3072 20000 > /tmp/long.txt & ) ; sleep 1.0; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | head; sleep 0.5; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|fclk\|mclk"| head -4; wait; sleep 3; cat /tmp/long.txt | grep -v "^layer"
