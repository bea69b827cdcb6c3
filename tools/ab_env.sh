# usage: ab_env.sh "A=1" "A=2 B=3" ...  (interleaved full-step bench runs, one environment per argument; "-" = default)
for r in 1 2; do for e in "$@"; do printf "%-28s " "$e"; if [ "$e" = "-" ]; then envs=""; else envs="$e"; fi; env $envs python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-inference 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pv=d['roofline']['per_variant']
print(d['value'], 'clips/s', d['ms_per_step'], 'ms |', ' '.join(f\"{k.split('<')[1][:-1]}={v['avg_ms']:.4f}\" for k,v in pv.items()))"; done; done
