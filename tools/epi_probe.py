import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
M = 100864
def run(tag, N, K, resid=True, scale=1.0, iters=7):
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda"); out = torch.empty((M, N), device="cuda")
    r = torch.randn((M, N), device="cuda") if resid else None
    for _ in range(2): ops.gemm(a, w, ops.EPI_F32, out, bias=bias, resid=r, scale=scale)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        e0.record(); ops.gemm(a, w, ops.EPI_F32, out, bias=bias, resid=r, scale=scale); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{tag:40s} N={N} K={K}: {sorted(ts)[len(ts)//2]:.3f} ms", flush=True)
for K in (64, 768, 3072):
    run("full (resid load + store)", 768, K)
    run("no resid", 768, K, resid=False)
    run("no store", 768, K, scale=-1.0)
    run("no resid load, no store", 768, K, resid=False, scale=-1.0)
