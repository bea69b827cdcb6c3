"""How fast is the vendor library (torch.mm -> hipBLASLt / rocBLAS) on the block's plain bf16 GEMMs, beside aim_gemm_bf16?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
M = 100864
def t(fn, n=7):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for name, N, K in (("qkv_fwd", 2304, 768), ("dxn (cat1 dgrad)", 768, 3264), ("qkv_dgrad", 768, 2304), ("out_dgrad", 768, 768), ("cfc", 3264, 768)):
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * M * N * K
    ms_a = t(lambda: ops.gemm(a, w, ops.EPI_BF16, out))
    wt = w.t().contiguous()
    ms_b = t(lambda: torch.mm(a, w.t(), out=out))
    ms_c = t(lambda: torch.mm(a, wt, out=out))
    print(f"{name:18s} N={N:5d} K={K:5d}: aim {ms_a:.3f} ms {fl/ms_a/1e9:7.1f} TF | torch.mm(a, w^T) {ms_b:.3f} ms {fl/ms_b/1e9:7.1f} TF | torch.mm(a, wT contiguous) {ms_c:.3f} ms {fl/ms_c/1e9:7.1f} TF")
