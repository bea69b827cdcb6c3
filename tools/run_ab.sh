for t in 256 3; do echo "== tile $t"; AIM_GEMM_TILE=$t python tools/bench_gemm.py 2>&1 | grep TFLOP | awk '{printf "%s %s | ", $1, $(NF-3)} END {print ""}'; done
