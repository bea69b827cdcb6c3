# usage: tools/ab3.sh  (GPU box): interleaved whole-step runs: base lib, new lib, new lib with the two-kernel attention backward
L=$PWD/tools/bin
run() { printf "%-12s " "$1"; shift; env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-inference 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'clips/s', d['ms_per_step'], 'ms')"; }
for r in 1 2 3; do
  run base AIM_HIP_LIB=$L/libaim_base.so
  run new AIM_HIP_LIB=$L/libaim_new.so
  run new-pipe AIM_HIP_LIB=$L/libaim_new.so AIM_ATTN_BWD_PIPE=0
done
