"""Reference point for the attention kernels: torch's scaled_dot_product_attention (whatever backend ROCm picks) on the
same problem (512 frames x 12 heads x 197 tokens x 64), forward and forward+backward, beside aim_attn_fwd / aim_attn_bwd."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
BT, N, H = 512, 197, 12
D = H * 64
def t(fn, n=7):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
qkv = torch.randn((BT * N, 3 * D), device="cuda").to(torch.bfloat16)
out = torch.empty((BT * N, D), dtype=torch.bfloat16, device="cuda")
lse = torch.empty((BT, H, N), device="cuda"); delta = torch.empty_like(lse)
do = torch.randn((BT * N, D), device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv)
print(f"aim_attn_fwd {t(lambda: ops.attn_fwd(qkv, out, lse, BT, N, H)):.3f} ms   aim_attn_bwd {t(lambda: ops.attn_bwd(qkv, out, do, lse, delta, dqkv, BT, N, H)):.3f} ms")
q, k, v = [x.reshape(BT, N, H, 64).permute(0, 2, 1, 3).contiguous().requires_grad_(True) for x in qkv.split(D, dim=1)]
g = torch.randn((BT, H, N, 64), device="cuda").to(torch.bfloat16)
for name, ctx in (("default", None),):
    try:
        fwd = t(lambda: F.scaled_dot_product_attention(q, k, v))
        def fb():
            o = F.scaled_dot_product_attention(q, k, v)
            o.backward(g)
        both = t(fb)
        print(f"torch SDPA ({name}, contiguous [B,H,N,64] inputs): fwd {fwd:.3f} ms   fwd+bwd {both:.3f} ms  (bwd ~ {both - fwd:.3f} ms)")
    except Exception as e:  # noqa: BLE001
        print("SDPA failed:", repr(e)[:200])
