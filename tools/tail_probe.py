"""What do the tile-round tails of the persistent GEMM cost?  The same launch at the workload's M and at the largest M below it
that fills whole rounds of 256 x 256 tiles on 256 CUs (stand-alone, HIP events, median of 9; chip warmed up first):
ViT-L/14 fp8 inference (12 views: M = 98 688), ViT-L/14 training (M = 131 584), ViT-B/16 training (M = 100 864)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
dev = "cuda"


def t(fn, n=9):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2] * 1e3


def whole_rounds(M, N, cus=256):
    tn = (N + 255) // 256
    tiles = ((M + 255) // 256) * tn
    full = tiles // cus * cus
    return full // tn * 256, tiles / cus


def bf16_case(tag, M, N, K, epi):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty((M, N), device=dev, dtype=torch.float32 if epi == ops.EPI_F32 else torch.bfloat16)
    resid = torch.randn((M, N), device=dev) if epi == ops.EPI_F32 else None
    M0, rounds = whole_rounds(M, N)
    full = t(lambda: ops.gemm(a, w, epi, out, resid=resid))
    cut = t(lambda: ops.gemm(a[:M0], w, epi, out[:M0], resid=None if resid is None else resid[:M0]))
    print(f"  {tag:34s} M={M} ({rounds:.2f} rounds) {full:7.1f} us | M={M0} {cut:7.1f} us | tail {full - cut:6.1f} us for {100 * (M - M0) / M:.2f} % of the rows")


def fp8_case(tag, M, N, K, epi):
    a8 = torch.randn((M, K), device=dev).to(ops.FP8)
    w8, sc = ops.quantize_fp8_rows(torch.randn((N, K), device=dev) * 0.02)
    if epi == ops.EPI_ACT8:
        out = torch.empty((M, N), device=dev, dtype=ops.FP8)
        kw = dict(act=ops.ACT_QGELU)
    else:
        out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        kw = dict(resid=torch.randn((M, N), device=dev).to(torch.bfloat16)) if epi == ops.EPI_RES16 else {}
    M0, rounds = whole_rounds(M, N)
    full = t(lambda: ops.gemm_fp8(a8, w8, sc, epi, out, **kw))
    kw0 = {k: (v[:M0] if k == "resid" else v) for k, v in kw.items()}
    cut = t(lambda: ops.gemm_fp8(a8[:M0], w8, sc, epi, out[:M0], **kw0))
    print(f"  {tag:34s} M={M} ({rounds:.2f} rounds) {full:7.1f} us | M={M0} {cut:7.1f} us | tail {full - cut:6.1f} us for {100 * (M - M0) / M:.2f} % of the rows")


x = torch.randn((8192, 8192), device=dev).to(torch.bfloat16)
for _ in range(30):
    x @ x
print("ViT-L/14 fp8 inference, 12 views x 32 frames (per layer):")
M = 12 * 32 * 257
fp8_case("QKV (N=3072, K=1024)", M, 3072, 1024, ops.EPI_BF16)
fp8_case("out_proj RES16 (N=1024, K=1024)", M, 1024, 1024, ops.EPI_RES16)
fp8_case("[c_fc|D_fc1] ACT8 (N=4352, K=1024)", M, 4352, 1024, ops.EPI_ACT8)
fp8_case("[c_proj|D_fc2] RES16 (N=1024,K=4352)", M, 1024, 4352, ops.EPI_RES16)
print("ViT-L/14 training forward, 32 clips x 16 frames (per layer):")
M = 32 * 16 * 257
bf16_case("QKV (N=3072, K=1024)", M, 3072, 1024, ops.EPI_BF16)
bf16_case("out_proj F32 (N=1024, K=1024)", M, 1024, 1024, ops.EPI_F32)
bf16_case("c_proj F32 (N=1024, K=4352)", M, 1024, 4352, ops.EPI_F32)
print("ViT-B/16 training, 64 clips x 8 frames (per layer):")
M = 64 * 8 * 197
bf16_case("[c_fc|D_fc1] as BF16 (N=3264, K=768)", M, 3264, 768, ops.EPI_BF16)
bf16_case("c_proj F32 (N=768, K=3264)", M, 768, 3264, ops.EPI_F32)
bf16_case("QKV (N=2304, K=768)", M, 2304, 768, ops.EPI_BF16)
