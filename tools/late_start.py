"""How much does a resident foreign kernel delay a persistent GEMM launched right behind it on another stream?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
dev = torch.device("cuda:0")
M = 100864
a = torch.randn((M, 768), device=dev).to(torch.bfloat16)
w = (torch.randn((768, 768), device=dev) * 0.03).to(torch.bfloat16)
out = torch.empty((M, 768), dtype=torch.bfloat16, device=dev)
dyb = torch.randn((M, 768), device=dev).to(torch.bfloat16)
a_s = torch.randn((M, 192), device=dev).to(torch.bfloat16)
w2, b2 = torch.zeros((768, 192), device=dev), torch.zeros(768, device=dev)
x32 = torch.randn((M, 768), device=dev); g = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
xl = torch.empty((M, 768), dtype=torch.bfloat16, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
def run(with_wgrad, with_ln, n=20):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sa):
            if with_ln:
                ops.layernorm_fwd(x32, g, b, M, 768, 768, y_bf16=xl, mean=mean, rstd=rstd)   # a non-persistent kernel first
            ev = torch.cuda.Event(); ev.record(sa)
        if with_wgrad:
            with torch.cuda.stream(sb):
                ops.wgrad(dyb, a_s, w2, b2)         # gets its workgroups in beside / behind the LayerNorm
        with torch.cuda.stream(sa):
            e0.record(sa)
            ops.gemm(a, w, ops.EPI_BF16, out)
            e1.record(sa)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]
for wl in (False, True):
    for ww in (False, True):
        print(f"LayerNorm before: {wl}  wgrad on the other stream: {ww}  ->  GEMM (N=K=768) {run(ww, wl) * 1e3:.0f} us")
