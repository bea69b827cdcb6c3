import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
def run(M, N, K, epi=ops.EPI_BF16, iters=9):
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16 if epi != ops.EPI_F32 else torch.float32, device="cuda")
    for _ in range(3): ops.gemm(a, w, epi, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        e0.record(); ops.gemm(a, w, epi, out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts)//2]
    tiles = (M // 256) * ((N + 255) // 256)
    print(f"M={M:7d} N={N:5d} K={K:5d} tiles/CU={tiles/256:5.2f}: {ms*1e3:8.1f} us   {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
for K in (64, 768, 3072):
    for rounds in (1, 2, 4, 8):
        run(256 * 256 * rounds, 256, K)
run(256*256*4, 256, 768, ops.EPI_F32)
# empty kernel launch overhead reference
x = torch.empty(1024, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): x.fill_(1)
e0.record(); x.fill_(1); e1.record(); torch.cuda.synchronize(); print("tiny fill kernel:", e0.elapsed_time(e1)*1e3, "us")
