import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
BT, N, H = 512, 197, 12
D = H * 64
qkv = torch.randn((BT * N, 3 * D), device="cuda").to(torch.bfloat16)
out = torch.empty((BT * N, D), dtype=torch.bfloat16, device="cuda")
lse = torch.empty((BT, H, N), device="cuda"); delta = torch.empty_like(lse)
do = torch.randn((BT * N, D), device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv)
def t(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
f = 4.0 * N * N * 64 * BT * H
ms = t(lambda: ops.attn_fwd(qkv, out, lse, BT, N, H)); print(f"attn_fwd {ms:.3f} ms  {f/ms/1e9:.1f} TFLOP/s")
ms = t(lambda: ops.attn_bwd(qkv, out, do, lse, delta, dqkv, BT, N, H)); print(f"attn_bwd {ms:.3f} ms  {2.5*f/ms/1e9:.1f} TFLOP/s (5 products)")
