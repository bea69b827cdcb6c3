# usage: tools/ab_lib_sec.sh LIB_A LIB_B ...   interleaved runs of the ViT-L/14 secondary workload (bench.py --secondary-only)
run() { printf "%-14s " "$1"; if [ "$1" = default ]; then E=""; else E="AIM_HIP_LIB=tools/bin/libaim_$1.so"; fi; env $E python bench.py --secondary-only 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'clips/s', d['ms_per_step'], 'ms')"; }
for r in 1 2 3; do for v in "$@"; do run "$v"; done; done
