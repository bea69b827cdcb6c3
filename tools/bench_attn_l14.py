"""Stand-alone times of the N = 257 (ViT-L/14) attention kernels: 512 frames x 16 heads, per kernel (HIP events around the
whole call; the two-kernel backward's split comes from `rocprofv3 --kernel-trace --stats -- python3 tools/bench_attn_l14.py`)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
BT, N, H = 512, 257, 16
D = H * 64
qkv = torch.randn((BT * N, 3 * D), device="cuda").to(torch.bfloat16)
out = torch.empty((BT * N, D), dtype=torch.bfloat16, device="cuda")
lse = torch.empty((BT, H, N), device="cuda"); delta = torch.empty_like(lse)
do = torch.randn((BT * N, D), device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv)
def t(fn, n=9):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
print(os.environ.get("AIM_HIP_LIB", "default"))
ms = t(lambda: ops.attn_fwd(qkv, out, lse, BT, N, H)); print(f"  attn_fwd {ms*1e3:.0f} us")
ms = t(lambda: ops.attn_bwd(qkv, out, do, lse, delta, dqkv, BT, N, H)); print(f"  attn_bwd (dq + dkv) {ms*1e3:.0f} us")
