# usage: ab_trees.sh DIR_A DIR_B ...  (interleaved full-step bench runs of several checkouts of this repo, one gpurun call)
for r in 1 2 3; do for d in "$@"; do printf "%-12s " "$d"; (cd $d && python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-inference 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pv=d['roofline']['per_variant']
print(d['value'], 'clips/s', d['ms_per_step'], 'ms |', ' '.join(f\"{k.split('<')[1][:-1]}={v['avg_ms']:.4f}\" for k,v in pv.items()))"); done; done
