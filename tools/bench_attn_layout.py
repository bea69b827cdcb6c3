"""Does the head-interleaved [M, 3*H*64] layout cost the attention kernels bandwidth?  Same work with H=1 (row stride 384 B)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
def t(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for BT, H in ((512, 12), (512 * 12, 1)):
    N, D = 197, H * 64
    qkv = torch.randn((BT * N, 3 * D), device="cuda").to(torch.bfloat16)
    out = torch.empty((BT * N, D), dtype=torch.bfloat16, device="cuda")
    lse = torch.empty((BT, H, N), device="cuda"); delta = torch.empty_like(lse)
    do = torch.randn((BT * N, D), device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv)
    print(f"BT={BT} H={H}: fwd {t(lambda: ops.attn_fwd(qkv, out, lse, BT, N, H)):.3f} ms  bwd {t(lambda: ops.attn_bwd(qkv, out, do, lse, delta, dqkv, BT, N, H)):.3f} ms")
