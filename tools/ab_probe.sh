# usage: ab_probe.sh libA libB ...  (interleaved rounds of tools/probe_gemm.py: K-loop and epilogue us per tile)
L=$PWD/adapt-image-models_amd
for r in 1 2 3; do for v in "$@"; do printf "%-10s " $v; AIM_HIP_LIB=$L/libaim_$v.so python tools/probe_gemm.py 2>&1 | grep tiles | awk '{printf "%s K %s E %s | ", $1, $14, $23} END {print ""}'; done; done
