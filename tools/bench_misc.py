import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
M, N = 100864, 197
def t(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2] * 1e3
X = torch.randn((M, 768), device="cuda").to(torch.bfloat16); o = torch.zeros(768, device="cuda"); at = torch.rand(N, device="cuda")
print(f"colsum 768 +at : {t(lambda: ops.colsum(X, o, at=at, ntok=N)):.1f} us  (155 MB)")
X2 = torch.randn((M, 3264), device="cuda").to(torch.bfloat16)[:, 3072:]; o2 = torch.zeros(192, device="cuda")
print(f"colsum 192 strided: {t(lambda: ops.colsum(X2, o2)):.1f} us  (39 MB)")
g = torch.randn((M, 768), device="cuda").to(torch.bfloat16); a = torch.randn((M, 192), device="cuda").to(torch.bfloat16)
dw = torch.zeros((768, 192), device="cuda"); dw2 = torch.zeros((192, 768), device="cuda")
print(f"wgrad 768x192 : {t(lambda: ops.wgrad(g, a, dw)):.1f} us")
print(f"wgrad 192x768 (strided G): {t(lambda: ops.wgrad(X2, g, dw2)):.1f} us")
x = torch.randn((M, 768), device="cuda"); fs = torch.zeros((512, 768), device="cuda")
print(f"frame_sum: {t(lambda: ops.frame_sum(x, at, fs, 512, N, 768)):.1f} us (310 MB)")
gam = torch.ones(768, device="cuda"); mean = torch.zeros(M, device="cuda"); rstd = torch.ones(M, device="cuda")
yb = torch.empty((M, 768), dtype=torch.bfloat16, device="cuda"); dx = torch.empty_like(x); dres = torch.randn_like(x)
print(f"ln_fwd: {t(lambda: ops.layernorm_fwd(x, gam, gam, M, 768, 768, y_bf16=yb, mean=mean, rstd=rstd)):.1f} us (465 MB)")
print(f"ln_bwd (bf16 dy): {t(lambda: ops.layernorm_bwd(g, x, gam, mean, rstd, M, 768, lddy=768, ldx=768, lddx=768, dres=dres, dx=dx, dx_bf16=yb)):.1f} us (1.24 GB)")
