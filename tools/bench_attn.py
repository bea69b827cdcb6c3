import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
BT, N, H = 512, 197, 12
D = H * 64
qkv = torch.randn((BT * N, 3 * D), device="cuda").to(torch.bfloat16)
out = torch.empty((BT * N, D), dtype=torch.bfloat16, device="cuda")
lse = torch.empty((BT, H, N), device="cuda"); delta = torch.empty_like(lse)
do = torch.randn((BT * N, D), device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv)
def t(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
f = 4.0 * N * N * 64 * BT * H
ms = t(lambda: ops.attn_fwd(qkv, out, lse, BT, N, H)); print(f"attn_fwd {ms:.3f} ms  {f/ms/1e9:.1f} TFLOP/s")
ms = t(lambda: ops.attn_bwd(qkv, out, do, lse, delta, dqkv, BT, N, H)); print(f"attn_bwd {ms:.3f} ms  {2.5*f/ms/1e9:.1f} TFLOP/s (5 products)")
if os.environ.get("STAMPS"):     # diagnostic build (AIM_HIP_LIB=libaim_stamps.so): per-workgroup 100 MHz time stamps in `delta`
    delta.zero_(); torch.cuda.synchronize()
    ops.attn_bwd(qkv, out, do, lse, delta, dqkv, BT, N, H); torch.cuda.synchronize()
    st = delta.reshape(-1)[:7 * 2 * 6 * 2].view(torch.int64).reshape(-1, 2, 6).cpu()
    names = ["prologue issue", "prologue wait+barrier", "first block", "remaining blocks", "epilogue"]
    for w, nm in ((0, "producer wave 0"), (1, "consumer wave 7")):
        d = (st[:, w, 1:] - st[:, w, :-1]).float().median(0).values * 0.01
        print(nm, " | ".join(f"{n} {float(x):.2f} us" for n, x in zip(names, d)), "| total", float((st[:, w, 5] - st[:, w, 0]).float().median()) * 0.01)

if os.environ.get("STAMPS") == "2":  # pipelined kernel: ticks 8..15 of workgroup 0, waves 0 (producer) and 7 (consumer)
    delta.zero_(); torch.cuda.synchronize()
    ops.attn_bwd(qkv, out, do, lse, delta, dqkv, BT, N, H); torch.cuda.synchronize()
    st = delta.reshape(-1)[:160].view(torch.int64).reshape(2, 8, 5).cpu()
    for w, nm in ((0, "producer wave 0"), (1, "consumer wave 7")):
        print(nm)
        for a in range(8):
            r = st[w, a]
            nxt = st[w, a + 1, 0] if a < 7 else r[4]
            print(f"  tick {8 + a}: wait {(r[1]-r[0])*0.01:.2f} barrier {(r[2]-r[1])*0.01:.2f} issue {(r[3]-r[2])*0.01:.2f} compute {(r[4]-r[3])*0.01:.2f} | tick {(nxt-r[0])*0.01:.2f} us")
