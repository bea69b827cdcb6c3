#!/usr/bin/env python3
"""Per-tile epilogue time of the fused [W_fc ; D_fc1] GEMM (N = 3072 + 192): QuickGELU column tiles vs the adapter's erf-GELU tile."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
M, N, K, H4 = 100864, 3264, 768, 3072
a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
at = torch.rand(197, device="cuda")
for epi, nm in ((ops.EPI_ACT, "ACT"), (ops.EPI_DACT, "DACT")):
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    kw = dict(act=ops.ACT_QGELU, n_split=H4, act2=ops.ACT_GELU, at=at, ntok=197, aux_grad=bool(int(os.environ.get("AUXG", "1"))))
    frag = bool(int(os.environ.get("FRAG", "1")))
    kw["aux_frag"] = frag
    side = ops.frag_buffer(M, N, "cuda").normal_() if frag else torch.randn((M, N), device="cuda").to(torch.bfloat16)
    if epi == ops.EPI_ACT:
        kw.update(bias=bias, out2=side)
    else:
        kw.update(aux=side)
    for _ in range(2):
        ops.gemm(a, w, epi, out, **kw)
    tiles = 394 * 13
    buf = torch.zeros((tiles + 512, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ops.gemm(a, w, epi, out, probe=buf, **kw)
    torch.cuda.synchronize()
    p = buf.cpu().numpy()
    p = p[p[:, 1] > 0]
    ep = (p[:, 3] - p[:, 2]) * 0.01
    srt = np.sort(ep)
    print(f"{nm}: tiles {len(p)}  epilogue us: median {np.median(ep):.2f}  p90 {srt[int(0.9 * len(ep))]:.2f}  p95 {srt[int(0.95 * len(ep))]:.2f} "
          f"p99 {srt[int(0.99 * len(ep))]:.2f} max {srt[-1]:.2f};  slowest 7.7 % mean {srt[int(0.923 * len(ep)):].mean():.2f}")
