import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
frames, ntok, D = 512, 197, 768
M = frames * ntok
dev = "cuda"
x = torch.randn((M, D), device=dev); g = torch.ones(D, device=dev)
dy = torch.randn((M, D), device=dev).to(torch.bfloat16); dres = torch.randn((M, D), device=dev).to(torch.bfloat16)
mean = x.mean(1).contiguous(); rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
out = torch.empty((M, D), dtype=torch.bfloat16, device=dev); w = torch.rand(ntok, device=dev)
part = torch.empty((frames, ops.LN_FSUM_GROUPS, D), device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("ln_bwd      %.1f us" % t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, M, D, lddy=D, ldx=D, lddx=D, dres=dres, dx_bf16=out)))
for G in (13, 25, 33, 50):
    ops.LN_FSUM_GROUPS = G
    part = torch.empty((frames, G, D), device=dev)
    print("ln_bwd_fsum G=%d %.1f us" % (G, t(lambda: ops.layernorm_bwd_fsum(dy, x, g, mean, rstd, dres, out, w, part, frames, ntok, D))))
