# usage: tools/ab_lib.sh LIB_A LIB_B ...   (GPU box): interleaved whole-step runs, one library build per variant
# ("default" = the in-tree library, otherwise tools/bin/libaim_NAME.so)
run() { printf "%-14s " "$1"; if [ "$1" = default ]; then E=""; else E="AIM_HIP_LIB=tools/bin/libaim_$1.so"; fi; env $E python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-inference 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'clips/s', d['ms_per_step'], 'ms')"; }
for r in 1 2 3; do for v in "$@"; do run "$v"; done; done
