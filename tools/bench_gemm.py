#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernels on the AIM block's shapes (one process, interleaved rounds)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops  # noqa: E402

DEV = "cuda"
M = int(os.environ.get("M", 100864))
SHAPES = [  # (name, N, K, epi)
    ("qkv_fwd", 2304, 768, ops.EPI_BF16), ("out_fwd", 768, 768, ops.EPI_F32), ("cfc_fwd", 3072, 768, ops.EPI_ACT),
    ("cproj_fwd", 768, 3072, ops.EPI_F32), ("cproj_dgrad", 3072, 768, ops.EPI_DACT), ("cfc_dgrad", 768, 3072, ops.EPI_F32),
    ("qkv_dgrad", 768, 2304, ops.EPI_F32), ("ad1_fwd", 192, 768, ops.EPI_ACT), ("ad2_fwd", 768, 192, ops.EPI_F32),
]


def run(name, N, K, epi, iters=5):
    a = torch.randn((M, K), device=DEV).to(torch.bfloat16)
    w = (torch.randn((N, K), device=DEV) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    kw = dict(bias=bias)
    if epi == ops.EPI_F32:
        out = torch.empty((M, N), device=DEV)
        kw["resid"] = torch.randn((M, N), device=DEV)
    else:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    if epi == ops.EPI_ACT:
        kw["out2"] = torch.empty_like(out)
    if epi == ops.EPI_DACT:
        kw["aux"] = torch.randn((M, N), device=DEV).to(torch.bfloat16)
    for _ in range(2):
        ops.gemm(a, w, epi, out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        e0.record()
        ops.gemm(a, w, epi, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    print(f"{name:12s} M={M} N={N:5d} K={K:5d}  {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    print("tile:", os.environ.get("AIM_GEMM_TILE", "256"))
    for s in SHAPES:
        run(*s)
