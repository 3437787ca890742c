#!/bin/bash
# Package power / sclk while a gemm_lab probe build loops (diagnostic).
# usage: power_probe.sh "<-D flags>" [reps]     -> one line: flags, ms per bench step, median sclk, median W
F="$1"; REPS=${2:-1500}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $F "$(dirname "$0")/gemm_lab.hip" -o /tmp/gemm_lab_p || exit 1
( LAB_ONE=1 LAB_REPS=$REPS /tmp/gemm_lab_p 0 136 > /tmp/lab_out.txt ) &
LP=$!
sleep 2.0
S=$(for i in 1 2 3 4 5 6 7; do kill -0 $LP 2>/dev/null || break; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed "s/.*: //" | tr "\n" " "; echo; sleep 0.25; done | awk '{gsub(/[()Mhz]/,"",$1); print $1, $NF}' | sort -k2 -n | awk '{a[NR]=$0} END {print a[int((NR+1)/2)]}')
wait $LP
echo "$F | $(grep '=>' /tmp/lab_out.txt | tail -1 | sed 's/.*=> //; s/ per bench.*//') | sclk_MHz W: $S"
