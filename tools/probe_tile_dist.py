#!/usr/bin/env python3
"""Per-tile K-loop / epilogue time distribution of the step's GEMM launches with their real epilogue arguments
(row factors, per-frame vectors, residual): looks for straggler tiles."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
M, D, NT = 100864, 768, 197
dev = "cuda"
at = (torch.rand(NT, device=dev) > 0.2).float() * 1.25
BT = M // NT


def probe(name, a, w, epi, out, **kw):
    for _ in range(2):
        ops.gemm(a, w, epi, out, **kw)
    tiles = ((a.shape[0] + 255) // 256) * ((w.shape[0] + 255) // 256)
    buf = torch.zeros((tiles + 512, 4), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ops.gemm(a, w, epi, out, probe=buf, **kw)
    torch.cuda.synchronize()
    p = buf.cpu().numpy()
    p = p[p[:, 1] > 0]
    kl = (p[:, 2] - p[:, 1]) * 0.01
    ep = (p[:, 3] - p[:, 2]) * 0.01
    span = (p[:, 3].max() - p[:, 1].min()) * 0.01
    q = lambda v, f: np.sort(v)[int(f * (len(v) - 1))]
    print(f"{name:22s} span {span:7.1f} us | K-loop p50 {q(kl, .5):6.2f} p99 {q(kl, .99):6.2f} max {kl.max():6.2f} | "
          f"epilogue p50 {q(ep, .5):6.2f} p90 {q(ep, .9):6.2f} p99 {q(ep, .99):6.2f} max {ep.max():6.2f}", flush=True)


def r(shape, dt=torch.bfloat16):
    return torch.randn(shape, device=dev).to(dt)


x = torch.randn((M, D), device=dev)
# out_proj forward: F32, residual, (1 - lamda) per frame (af), S_Adapter vector per frame (vec) with DropPath factor (bt)
probe("out_proj F32 af/vec/bt", r((M, D)), r((D, D)) * D ** -0.5, ops.EPI_F32, torch.empty_like(x), bias=torch.randn(D, device=dev), resid=x,
      af=torch.rand(BT, device=dev), vec=torch.randn((BT, D), device=dev), bt=at, ntok=NT)
# c_proj forward: F32, K = 3264, residual, bias row vec with DropPath factor
probe("c_proj F32 vec/bt", r((M, 3264)), r((D, 3264)) * 3264 ** -0.5, ops.EPI_F32, torch.empty_like(x), bias=torch.randn(D, device=dev), resid=x,
      vec=torch.randn((1, D), device=dev), ldv=0, bt=at, ntok=NT)
probe("qkv BF16", r((M, D)), r((3 * D, D)) * D ** -0.5, ops.EPI_BF16, torch.empty((M, 3 * D), dtype=torch.bfloat16, device=dev), bias=torch.randn(3 * D, device=dev))
probe("dxn BF16 K=3264", r((M, 3264)), r((D, 3264)) * 3264 ** -0.5, ops.EPI_BF16, torch.empty((M, D), dtype=torch.bfloat16, device=dev))
probe("dao BF16 K=768", r((M, D)), r((D, D)) * D ** -0.5, ops.EPI_BF16, torch.empty((M, D), dtype=torch.bfloat16, device=dev))
