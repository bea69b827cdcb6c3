"""Stand-alone time of the adapter weight-gradient calls of ONE ViT-B/16 block at 64 clips (the in-step rocprof durations of
these kernels are stretched: they run on the detached stream in whatever CUs the persistent GEMMs leave)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
M, D, r, N = 100864, 768, 192, 197
dev = "cuda"
dyb = torch.randn((M, D), device=dev).to(torch.bfloat16)
hcat = torch.randn((M, 4 * D + r), device=dev).to(torch.bfloat16)
xn = torch.randn((M, D), device=dev).to(torch.bfloat16)
dcat = torch.randn((M, 4 * D + r), device=dev).to(torch.bfloat16)
at = torch.rand(N, device=dev)
w2, b2 = torch.zeros((D, r), device=dev), torch.zeros(D, device=dev)
w1, b1 = torch.zeros((r, D), device=dev), torch.zeros(r, device=dev)
small = [(torch.randn((512, D), device=dev).to(torch.bfloat16), torch.randn((512, r), device=dev).to(torch.bfloat16)) for _ in range(2)]


def block():
    ops.wgrad(dyb, hcat[:, 4 * D:], w2, b2, at=at, ntok=N)          # MLP_Adapter.D_fc2 (+ DropPath-scaled bias)
    ops.wgrad(dcat[:, 4 * D:], xn, w1, b1)                           # MLP_Adapter.D_fc1 (+ bias)
    for g_, a_ in small:                                             # S_/T_Adapter on B*T = 512 rows
        ops.wgrad(g_, a_, w2, b2)
        ops.wgrad(a_, g_, w1, b1)


for _ in range(3):
    block()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    block()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"weight gradients of one block, alone on the chip: {ms * 1e3:.0f} us (6 wgrad + 6 finish launches) -> {ms * 12:.2f} ms per 12-layer step")
