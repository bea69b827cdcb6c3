#!/usr/bin/env python3
"""Does the consumer of a just-written 310 MB tensor run faster when it starts with what the producer wrote LAST?
(Infinity Cache / MALL retention of writes.)  Writer: two half copies in either order; reader: ln_fwd over all rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
M, D = 100864, 768
src = torch.randn((M, D), device="cuda")
x = torch.empty_like(src)
gam = torch.ones(D, device="cuda"); mean = torch.zeros(M, device="cuda"); rstd = torch.ones(M, device="cuda")
yb = torch.empty((M, D), dtype=torch.bfloat16, device="cuda")
h = M // 2


def run(order, parts):
    ts = []
    for _ in range(7):
        bounds = [(i * M // parts, (i + 1) * M // parts) for i in range(parts)]
        if order == "last-first":
            bounds = bounds[::-1]
        for a, b in bounds:
            x[a:b].copy_(src[a:b])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.layernorm_fwd(x, gam, gam, M, D, D, y_bf16=yb, mean=mean, rstd=rstd)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for parts in (2, 4, 8):
    a = run("first-first", parts)     # producer wrote row block 0 first ... reader starts with the OLDEST data
    b = run("last-first", parts)      # producer wrote row block 0 last  ... reader starts with the NEWEST data
    print(f"{parts} parts: reader starts with oldest rows {a:.1f} us, with newest rows {b:.1f} us")
