#!/usr/bin/env python3
"""Per-tile timeline of the persistent 256x256 GEMM (aim_gemm_args.probe): K-loop vs epilogue time per tile."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops  # noqa: E402
from aim_amd.lib import load_library  # noqa: E402

M = int(os.environ.get("M", 100864))
SHAPES = [("qkv_fwd", 2304, 768, ops.EPI_BF16), ("out_fwd", 768, 768, ops.EPI_F32), ("cfc_fwd", 3072, 768, ops.EPI_ACT),
          ("cproj_fwd", 768, 3072, ops.EPI_F32), ("cproj_dgrad", 3072, 768, ops.EPI_DACT)]


def run(name, N, K, epi):
    dev = "cuda"
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
    kw = dict(bias=torch.randn(N, device=dev))
    if epi == ops.EPI_F32:
        out = torch.empty((M, N), device=dev)
        kw["resid"] = torch.randn((M, N), device=dev)
    else:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    if epi == ops.EPI_ACT:
        kw["out2"] = torch.empty_like(out)
    if epi == ops.EPI_DACT:
        kw["aux"] = torch.randn((M, N), device=dev).to(torch.bfloat16)
    for _ in range(2):
        ops.gemm(a, w, epi, out, **kw)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    cap = tiles + 512
    buf = torch.zeros((cap, 4), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ops.gemm(a, w, epi, out, probe=buf, **kw)
    torch.cuda.synchronize()
    p = buf.cpu().numpy()
    if os.environ.get("STAMPS"):        # diagnostic build only (-DAIM_X_STAMPS): phase stamps of workgroup 0, waves 0 and 4
        raw = p[cap - 32:].reshape(-1)
        names = []
        for ph in "abcd":
            names += [f"R{ph}", f"bar", f"M{ph}", f"bar"]
        for w, off in ((0, 0), (4, 64)):
            st = raw[off:off + 17].astype(np.int64)
            print(f"   wave {w} cycles: " + " ".join(f"{n}={int(x)}" for n, x in zip(names, np.diff(st))) + f" | iteration {int(st[16] - st[0])}")
        p = p[:cap - 32]
    p = p[p[:, 1] > 0]
    cyc = p[:, 0] >> 16
    p[:, 0] &= 0xffff
    t0 = p[:, 1].min()
    kl = (p[:, 2] - p[:, 1]) * 0.01
    ep = (p[:, 3] - p[:, 2]) * 0.01
    total = (p[:, 3].max() - t0) * 0.01
    clk = np.median(cyc / np.maximum(p[:, 2] - p[:, 1], 1)) * 0.1     # cycles per 10 ns -> GHz
    nk = (K + 63) // 64
    print(f"   in-kernel clock {clk:5.2f} GHz; K-loop {np.mean(cyc) / nk:7.0f} cycles per K-step (MFMA floor 2048)")
    print(f"{name:12s} N={N} K={K}: tiles {len(p)}  kernel span {total:7.1f} us | K-loop us mean {kl.mean():6.2f} "
          f"p10 {np.percentile(kl, 10):6.2f} p90 {np.percentile(kl, 90):6.2f} | epilogue us mean {ep.mean():6.2f} "
          f"p10 {np.percentile(ep, 10):6.2f} p90 {np.percentile(ep, 90):6.2f}", flush=True)
    # by tile ordinal inside a workgroup: K-loop time of a workgroup's first tile (nothing in flight before it) vs later ones
    order = np.lexsort((p[:, 1], p[:, 0]))
    ps = p[order]
    first = np.r_[True, ps[1:, 0] != ps[:-1, 0]]
    ordn = np.arange(len(ps)) - np.maximum.accumulate(np.where(first, np.arange(len(ps)), 0))
    klo = (ps[:, 2] - ps[:, 1]) * 0.01
    gap = np.r_[0, (ps[1:, 1] - ps[:-1, 3]) * 0.01]
    print("   K-loop us by tile ordinal:", " ".join(f"{o}:{klo[ordn == o].mean():.2f}" for o in range(min(6, ordn.max() + 1))),
          "| start spread of ordinal 3 (p5..p95) us:", [round(float(x), 1) for x in np.percentile((ps[ordn == min(3, ordn.max()), 1] - t0) * 0.01, [5, 50, 95])])
    # first workgroup's timeline
    wg0 = p[p[:, 0] == 0]
    wg0 = wg0[np.argsort(wg0[:, 1])]
    print("   wg0 (start, kloop, epi) us:", [(round((r[1] - t0) * 0.01, 1), round((r[2] - r[1]) * 0.01, 1), round((r[3] - r[2]) * 0.01, 1)) for r in wg0[:8]])


if __name__ == "__main__":
    for s in SHAPES:
        run(*s)
