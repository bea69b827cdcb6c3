#!/usr/bin/env python3
"""Phase stamps of one attention-forward workgroup (diagnostic build: -DAIM_X_STAMPS of attn_fwd.hip)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
from aim_amd.lib import load_library
BT, N, H = 512, 197, 12
D = H * 64
qkv = torch.randn((BT * N, 3 * D), device="cuda").to(torch.bfloat16)
out = torch.empty((BT * N, D), dtype=torch.bfloat16, device="cuda")
lse = torch.empty((BT, H, N), device="cuda")
for _ in range(3):
    ops.attn_fwd(qkv, out, lse, BT, N, H)
buf = torch.zeros((8, 16), dtype=torch.int64, device="cuda")
lib = load_library()
fn = lib.aim_attn_probe
fn.argtypes = [ctypes.c_void_p]
torch.cuda.synchronize()
fn(buf.data_ptr())
ops.attn_fwd(qkv, out, lse, BT, N, H)
torch.cuda.synchronize()
fn(None)
p = buf.cpu().numpy()
names = ["stage", "sync", "-> q0", "S", "softmax-a", "softmax-b", "PV", "store", "->q1", "S", "softmax-a", "softmax-b", "PV", "store", "end"]
for w in range(8):
    st = p[w]
    idx = [i for i in range(16) if st[i] > 0]
    print(f"wave {w}: " + " ".join(f"{names[i-1] if i-1 < len(names) else i}={int(st[i]-st[j])}" for j, i in zip(idx[:-1], idx[1:])) + f" | total {int(st[15]-st[0])}")
