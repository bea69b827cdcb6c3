# usage: ab_bench.sh libA libB ...  (interleaved full-step bench runs in one gpurun call; prints clips/s and per-variant GEMM ms)
L=$PWD/adapt-image-models_amd
for r in 1 2; do for v in "$@"; do printf "%-10s " $v; AIM_HIP_LIB=$L/libaim_$v.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-inference 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pv=d['roofline']['per_variant']
print(d['value'], 'clips/s', d['ms_per_step'], 'ms |', ' '.join(f\"{k.split('<')[1][:-1]}={v['avg_ms']:.4f}\" for k,v in pv.items()))"; done; done
