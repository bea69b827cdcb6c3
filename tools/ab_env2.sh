# usage: tools/ab_env2.sh VAR=a VAR=b ...  (GPU box): interleaved whole-step runs, one env assignment per variant
run() { printf "%-22s " "$1"; env "$1" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-inference 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'clips/s', d['ms_per_step'], 'ms')"; }
for r in 1 2 3; do for v in "$@"; do run "$v"; done; done
