for g in 1 2 4 0; do echo "== groups $g (0 = auto)"; AIM_GEMM_GROUPS=$g python tools/bench_gemm.py 2>&1 | grep TFLOP | awk '{printf "%s %s | ", $1, $(NF-3)} END {print ""}'; done
