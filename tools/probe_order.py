import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
M, D = 100864, 768
dev = "cuda"
def t(fn, n=7):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2] * 1e3
def mk(scale_in_f32):
    a = torch.randn((M, D), device=dev).to(torch.bfloat16)
    if scale_in_f32:
        w = (torch.randn((3 * D, D), device=dev) * D ** -0.5).to(torch.bfloat16)
    else:
        w = torch.randn((3 * D, D), device=dev).to(torch.bfloat16) * D ** -0.5
    out = torch.empty((M, 3 * D), dtype=torch.bfloat16, device=dev)
    bias = torch.randn(3 * D, device=dev)
    return a, w, out, bias
for tag, f32 in (("w scaled in f32", True), ("w scaled in bf16", False), ("w scaled in f32", True)):
    a, w, out, bias = mk(f32)
    print(f"{tag}: qkv {t(lambda: ops.gemm(a, w, ops.EPI_BF16, out, bias=bias)):.1f} us  |w| mean {w.float().abs().mean().item():.4f}", flush=True)
# data dependence: zeros / small / large magnitudes
a, w, out, bias = mk(True)
for tag, aa, ww in (("a = 0", torch.zeros_like(a), w), ("a x 0.01", a * 0.01, w), ("a x 1", a, w), ("a x 8", a * 8, w), ("w x 30", a, w * 30), ("a const 1", torch.ones_like(a), w)):
    print(f"{tag}: qkv {t(lambda: ops.gemm(aa, ww, ops.EPI_BF16, out, bias=bias)):.1f} us", flush=True)
