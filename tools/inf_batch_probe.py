import os, sys, time, json
sys.path.insert(0, '/root/repo')
import torch, bench
from aim_amd import backbone as bb
dev = torch.device('cuda', 0)
for samples in (1, 4):
    r = bench.inference_l14(dev, 0, 1, samples=samples, steps=6, warmup=2)
    print(samples, 'samples x 3 views:', r['value'], 'views/s fp8', r['ms_per_step'], 'ms ; bf16', r['bf16_same_process'])
    print('   join stalls (ms over 16 steps):', {k: round(v, 2) for k, v in bb.join_stats().items()})
