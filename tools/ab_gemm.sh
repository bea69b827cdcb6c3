# usage: ab_gemm.sh libA libB ...   (interleaved rounds of tools/bench_gemm.py, one line per run)
L=$PWD/adapt-image-models_amd
for r in 1 2 3; do for v in "$@"; do printf "%-10s " $v; AIM_HIP_LIB=$L/libaim_$v.so python tools/bench_gemm.py 2>&1 | grep TFLOP | awk '{printf "%s %s | ", $1, $(NF-3)} END {print ""}'; done; done
