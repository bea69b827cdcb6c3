# usage: tools/ab_wg.sh (GPU box): weight-gradient workgroup count: stand-alone block time and whole step, interleaved
run() { printf "%-8s " "$1"; shift; env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-inference 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'clips/s', d['ms_per_step'], 'ms')"; }
for w in 256 192 128 96; do printf "wgs=%s: " $w; AIM_WGRAD_WGS=$w python tools/wgrad_only.py 2>&1 | tail -1; done
for r in 1 2; do for w in 256 192 128 96; do run wgs=$w AIM_WGRAD_WGS=$w; done; done
