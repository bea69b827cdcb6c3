"""Stand-alone time of each weight-gradient call of one block (wgrad + finish launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
M, D, r, N = 100864, 768, 192, 197
dev = "cuda"
dyb = torch.randn((M, D), device=dev).to(torch.bfloat16)
a_s = torch.randn((M, r), device=dev).to(torch.bfloat16)
xn = torch.randn((M, D), device=dev).to(torch.bfloat16)
da = torch.randn((M, r), device=dev).to(torch.bfloat16)
at = torch.rand(N, device=dev)
w2, b2 = torch.zeros((D, r), device=dev), torch.zeros(D, device=dev)
w1, b1 = torch.zeros((r, D), device=dev), torch.zeros(r, device=dev)
g5, a5 = torch.randn((512, D), device=dev).to(torch.bfloat16), torch.randn((512, r), device=dev).to(torch.bfloat16)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("D_fc2 [768x192] M=100864 (+bias, at): %.1f us" % t(lambda: ops.wgrad(dyb, a_s, w2, b2, at=at, ntok=N)))
print("D_fc1 [192x768] M=100864 (+bias):     %.1f us" % t(lambda: ops.wgrad(da, xn, w1, b1)))
print("D_fc2 [768x192] M=512:                %.1f us" % t(lambda: ops.wgrad(g5, a5, w2, b2)))
print("D_fc1 [192x768] M=512:                %.1f us" % t(lambda: ops.wgrad(a5, g5, w1, b1)))
