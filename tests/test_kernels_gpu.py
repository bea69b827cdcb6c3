"""Kernel-level numerics on a real MI355X: each HIP kernel (called through the C ABI) against a plain
PyTorch fp32 restatement of the same op on the same bf16-rounded inputs.  Tolerances are written
next to each check: bf16 outputs are allowed one bf16 ulp (2^-8 relative) plus accumulation noise."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from aim_amd import ops
    return ops


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(DEV)


def close(got, ref, atol, rtol, what=""):
    got, ref = got.float(), ref.float()
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = err > bound
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements off; max err {err.max().item():.4e} "
                           f"(ref max {ref.abs().max().item():.3e}) at {torch.nonzero(bad)[0].tolist()}")


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


# ------------------------------------------------------------------ GEMM -----------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (20, 32, 128), (100, 192, 136),
                                   (394, 2304, 768), (130, 132, 72), (1, 4, 8),
                                   # M >= 1024 runs the 256x256 pipelined kernel (gemm256.hip)
                                   (1024, 256, 64), (2048, 768, 768), (1300, 200, 136), (1576, 3072, 192),
                                   (1100, 128, 8), (3940, 192, 768), (1182, 768, 3072),
                                   # more tiles than CUs: balanced persistent grids that are not multiples of 8
                                   # (300 tiles -> 150 workgroups; 903 tiles -> 226), several tiles per workgroup
                                   (76800, 256, 64), (77000, 768, 72)])
def test_gemm_bf16_plain(M, N, K):
    ops = _ops()
    a, w = rnd((M, K), 1, dtype=torch.bfloat16), rnd((N, K), 2, K ** -0.5, torch.bfloat16)
    bias = rnd((N,), 3)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm(a, w, ops.EPI_BF16, out, bias=bias)
    ref = a.float() @ w.float().T + bias
    close(out, ref, 2e-3, 1e-2, f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("frames", [6, 260])
def test_gemm_strided_and_rowscale(frames):
    ops = _ops()
    ntok, N, K = 5, 192, 128
    M = ntok * frames
    big = rnd((M, 3 * K), 4, dtype=torch.bfloat16)
    a = big[:, K:2 * K]                      # row stride 3K, column offset K
    w = rnd((N, K), 5, K ** -0.5, torch.bfloat16)
    af, at = rnd((frames,), 6), rnd((ntok,), 7)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    ops.gemm(a, w, ops.EPI_BF16, out, af=af, at=at, ntok=ntok)
    rs = (af[:, None] * at[None, :]).reshape(M, 1)
    close(out, rs * (a.float() @ w.float().T), 2e-3, 1e-2, "gemm strided+rowscale")


@pytest.mark.parametrize("M", [200, 1500])
@pytest.mark.parametrize("act", [0, 1])
def test_gemm_act_and_dact(act, M):
    ops = _ops()
    N, K = 192, 128
    a, w = rnd((M, K), 8, dtype=torch.bfloat16), rnd((N, K), 9, K ** -0.5, torch.bfloat16)
    bias = rnd((N,), 10, 0.1)
    post = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    pre = torch.zeros_like(post)
    ops.gemm(a, w, ops.EPI_ACT, post, bias=bias, out2=pre, act=act)
    pre_ref = a.float() @ w.float().T + bias
    close(pre, pre_ref, 2e-3, 1e-2, "act pre")
    f = quick_gelu if act == 0 else torch.nn.functional.gelu
    close(post, f(pre.float()), 2e-3, 1e-2, "act post")       # applied to the bf16-rounded pre
    # DACT: out = (g @ w2.T) * act'(pre)
    g, w2 = rnd((M, K), 11, dtype=torch.bfloat16), rnd((N, K), 12, K ** -0.5, torch.bfloat16)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    ops.gemm(g, w2, ops.EPI_DACT, out, aux=pre, act=act)
    x = pre.float().requires_grad_(True)
    f(x).sum().backward()
    close(out, (g.float() @ w2.float().T) * x.grad, 3e-3, 1.5e-2, "dact")
    # aux_grad: the forward epilogue stores act'(bf16(pre)) instead of pre (same `post`), the dgrad epilogue multiplies by it
    post2 = torch.zeros_like(post)
    dsv = torch.zeros_like(post)
    ops.gemm(a, w, ops.EPI_ACT, post2, bias=bias, out2=dsv, act=act, aux_grad=True)
    assert torch.equal(post2, post)
    close(dsv, x.grad, 2e-3, 1e-2, "stored activation derivative")
    out2 = torch.zeros_like(out)
    ops.gemm(g, w2, ops.EPI_DACT, out2, aux=dsv, act=act, aux_grad=True)
    close(out2, (g.float() @ w2.float().T) * dsv.float(), 3e-3, 1.5e-2, "dact from the stored derivative")
    close(out2, out, 6e-3, 2e-2, "dact: both forms")
    if M >= 1024:
        # aux_frag: the side buffer in the GEMM pair's own fragment order (large-tile kernel only) -- same results, bit for bit
        for ag in (False, True):
            fb = ops.frag_buffer(M, N, DEV)
            post3, out3 = torch.zeros_like(post), torch.zeros_like(out)
            ops.gemm(a, w, ops.EPI_ACT, post3, bias=bias, out2=fb, act=act, aux_grad=ag, aux_frag=True)
            ops.gemm(g, w2, ops.EPI_DACT, out3, aux=fb, act=act, aux_grad=ag, aux_frag=True)
            assert torch.equal(post3, post) and torch.equal(out3, out2 if ag else out)
    else:
        with pytest.raises(Exception):
            ops.gemm(a, w, ops.EPI_ACT, post2, bias=bias, out2=ops.frag_buffer(M, N, DEV), act=act, aux_frag=True)


@pytest.mark.parametrize("frames", [4, 200])
def test_gemm_f32_residual_forms(frames):
    ops = _ops()
    ntok, N, K = 7, 128, 192
    M = ntok * frames
    a, w = rnd((M, K), 13, dtype=torch.bfloat16), rnd((N, K), 14, K ** -0.5, torch.bfloat16)
    bias, resid = rnd((N,), 15), rnd((M, N), 16)
    af, at, bt_, vec = rnd((frames,), 17), rnd((ntok,), 18), rnd((ntok,), 19), rnd((frames, N), 20)
    acc = a.float() @ w.float().T
    rs = (af[:, None] * at[None, :]).reshape(M, 1)
    vterm = (bt_[None, :, None] * vec[:, None, :]).reshape(M, N)
    out = torch.zeros((M, N), device=DEV)
    ops.gemm(a, w, ops.EPI_F32, out, bias=bias, resid=resid, af=af, at=at, vec=vec, bt=bt_, ntok=ntok)
    close(out, resid + rs * (acc + bias) + vterm, 1e-4, 1e-4, "f32 full")
    ops.gemm(a, w, ops.EPI_F32, out, bias=bias, resid=resid, at=at, ntok=ntok, rs_bias_only=True)
    close(out, resid + acc + at.repeat(frames)[:, None] * bias, 1e-4, 1e-4, "f32 rs_bias_only")
    ops.gemm(a, w, ops.EPI_F32, out)
    close(out, acc, 1e-4, 1e-4, "f32 plain")
    # in-place accumulate (resid is out)
    ref2 = out.clone() + acc
    ops.gemm(a, w, ops.EPI_F32, out, resid=out)
    close(out, ref2, 1e-4, 1e-4, "f32 accumulate in place")


@pytest.mark.parametrize("N", [197, 257, 129, 64, 300])       # 257 / 129 / 300: tiles holding one valid row or column
def test_gemm_expsum_batched(N):
    ops = _ops()
    BT, D = 3, 256
    qkv = rnd((BT * N, 3 * D), 21, 0.5, torch.bfloat16)
    nt = ops.expsum_tiles(N, N)
    part = torch.zeros((BT, nt, 2), device=DEV)
    q, k = qkv[:, :D], qkv[:, D:2 * D]
    ops.gemm(q, k, ops.EPI_EXPSUM, part, M=N, N=N, K=D, batch=BT, stride_a=N * 3 * D, stride_w=N * 3 * D,
             scale=0.125)
    s = torch.einsum("bik,bjk->bij", q.float().reshape(BT, N, D), k.float().reshape(BT, N, D)) * 0.125
    ref = torch.logsumexp(s.reshape(BT, -1), dim=1)
    got = torch.logsumexp(part[..., 0] + torch.log(part[..., 1]), dim=1)
    close(got, ref, 1e-4, 1e-5, "expsum logsumexp")


@pytest.mark.parametrize("frames,ntok,D", [(6, 197, 768), (3, 257, 1024), (5, 3, 64), (2, 50, 192)])
def test_layernorm_bwd_fsum(frames, ntok, D):
    """ln backward that also emits the per-frame weighted token sums of the dx it stores (the ln_2 backward of the block):
    dx equal to aim_layernorm_bwd's within one bf16 ulp, the finished sums equal to a frame_sum pass over that dx."""
    ops = _ops()
    M = frames * ntok
    x = rnd((M, D), 70, 2.0)
    g = rnd((D,), 71, 1.0) + 1.0
    dy = rnd((M, D), 72, 1.0, torch.bfloat16)
    dres = rnd((M, D), 73, 1.0, torch.bfloat16)
    w = torch.rand(ntok, device=DEV) * (torch.rand(ntok, device=DEV) > 0.3)       # DropPath-like factors, zeros included
    mean, rstd = x.mean(1).contiguous(), (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    ref = torch.empty((M, D), dtype=torch.bfloat16, device=DEV)
    ops.layernorm_bwd(dy, x, g, mean, rstd, M, D, lddy=D, ldx=D, lddx=D, dres=dres, dx_bf16=ref)
    want = torch.zeros((frames, D), device=DEV)
    ops.frame_sum(ref, w, want, frames, ntok, D)
    got_dx = torch.full((M, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    part = torch.full((frames, ops.LN_FSUM_GROUPS, D), float("nan"), device=DEV)
    ops.layernorm_bwd_fsum(dy, x, g, mean, rstd, dres, got_dx, w, part, frames, ntok, D)
    got = torch.zeros((frames, D), device=DEV)
    ops.frame_sum(part, None, got, frames, ops.LN_FSUM_GROUPS, D)
    # same arithmetic, separately compiled (fp contraction may differ in the last bit): at most one bf16 ulp apart
    close(got_dx, ref, 1e-6, 2 ** -7, "ln_bwd_fsum dx")
    # the sums are over the values THIS kernel stored (a last-bit difference in dx is a whole bf16 ulp in one term)
    direct = (got_dx.float().view(frames, ntok, D) * w.view(1, ntok, 1)).sum(1)
    close(got, want, 2e-3 * float(want.abs().max()) + 1e-6, 1e-5, "ln_bwd_fsum frame sums vs the two-pass form")
    close(got, direct, 1e-4 * float(direct.abs().max()) + 1e-6, 1e-5, "ln_bwd_fsum vs torch")


# ------------------------------------------------------------------ LayerNorm ------------------
@pytest.mark.parametrize("rows,D", [(7, 128), (1000, 768), (33, 1024)])
def test_layernorm_fwd_bwd(rows, D):
    ops = _ops()
    x = rnd((rows, D), 30, 2.0) + 0.5
    gamma, beta = rnd((D,), 31) * 0.1 + 1, rnd((D,), 32, 0.1)
    yb = torch.zeros((rows, D), dtype=torch.bfloat16, device=DEV)
    yf = torch.zeros((rows, D), device=DEV)
    mean, rstd = torch.zeros(rows, device=DEV), torch.zeros(rows, device=DEV)
    ops.layernorm_fwd(x, gamma, beta, rows, D, D, y_bf16=yb, y_f32=yf, mean=mean, rstd=rstd)
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    close(yf, ref, 2e-6, 2e-6, "ln fwd f32")
    close(yb, ref, 1e-6, 2 ** -8, "ln fwd bf16")
    dy, dres = rnd((rows, D), 33), rnd((rows, D), 34)
    ref.backward(dy)
    dx = torch.zeros((rows, D), device=DEV)
    dxb = torch.zeros((rows, D), dtype=torch.bfloat16, device=DEV)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, rows, D, lddy=D, ldx=D, lddx=D, dres=dres, dx=dx, dx_bf16=dxb,
                      dgamma=dg, dbeta=db)
    close(dx, xr.grad + dres, 2e-5, 1e-5, "ln bwd dx")
    close(dxb, xr.grad + dres, 1e-5, 2 ** -8, "ln bwd dx bf16")
    close(dg, gr.grad, 1e-3, 1e-4, "ln dgamma")
    close(db, br.grad, 1e-3, 1e-4, "ln dbeta")


def test_layernorm_strided_rows():
    """ln_post reads only the class rows: row stride N*D."""
    ops = _ops()
    BT, N, D = 6, 5, 128
    x = rnd((BT, N, D), 35)
    gamma, beta = rnd((D,), 36) + 1, rnd((D,), 37)
    y = torch.zeros((BT, D), device=DEV)
    ops.layernorm_fwd(x, gamma, beta, BT, D, N * D, y_f32=y)
    close(y, torch.nn.functional.layer_norm(x[:, 0], (D,), gamma, beta, 1e-5), 2e-6, 2e-6, "ln strided")


# ------------------------------------------------------------------ attention ------------------
def _attn_ref(qkv, BT, N, H):
    D = H * 64
    q, k, v = [t.reshape(BT, N, H, 64).permute(0, 2, 1, 3) for t in qkv.float().split(D, dim=1)]
    s = q @ k.transpose(-1, -2) / 8.0
    p = s.softmax(-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(BT * N, D)
    return o, torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("BT,N,H", [(2, 5, 2), (3, 50, 1), (4, 197, 12), (2, 257, 4), (1, 16, 1),
                                    (2, 64, 2), (2, 100, 3), (1, 224, 2), (3, 196, 2), (2, 65, 1)])   # 65 <= N <= 224: the pipelined fused backward
def test_attn_fwd_bwd(BT, N, H):
    ops = _ops()
    D = H * 64
    qkv = rnd((BT * N, 3 * D), 40, 1.0, torch.bfloat16)
    out = torch.full((BT * N, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros((BT, H, N), device=DEV)
    ops.attn_fwd(qkv, out, lse, BT, N, H)
    x = qkv.float().requires_grad_(True)
    ref, lse_ref = _attn_ref(x, BT, N, H)
    close(lse, lse_ref, 2e-3, 1e-4, "attn lse")
    close(out, ref, 6e-3, 1.5e-2, "attn out")     # P is rounded to bf16 before P@V
    do = rnd((BT * N, D), 41, 1.0, torch.bfloat16)
    ref.backward(do.float())
    dqkv = torch.full((BT * N, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
    delta = torch.zeros((BT, H, N), device=DEV)
    ops.attn_bwd(qkv, out, do, lse, delta, dqkv, BT, N, H)
    g = x.grad
    scale = g.abs().max().item()
    close(dqkv, g, 2e-2 * scale, 3e-2, "attn dqkv")


@pytest.mark.parametrize("BT,N,H", [(3, 197, 12), (2, 65, 1), (2, 224, 2), (5, 130, 3)])
def test_attn_bwd_two_kernel_form(BT, N, H, monkeypatch):
    """65 <= N <= 224 runs the pipelined fused backward by default; the two-kernel form (AIM_ATTN_BWD_PIPE=0: what N = 257 and
    the tiny shapes always run) is checked at these sizes too, against plain PyTorch autograd.  The switch is read once per
    process, so it is driven through a second copy of the library loaded with the switch set."""
    import ctypes, shutil, tempfile, os
    from aim_amd import lib as L
    ops = _ops()
    D = H * 64
    qkv = rnd((BT * N, 3 * D), 50, 1.0, torch.bfloat16)
    out = torch.empty((BT * N, D), dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros((BT, H, N), device=DEV)
    ops.attn_fwd(qkv, out, lse, BT, N, H)
    do = rnd((BT * N, D), 51, 1.0, torch.bfloat16)
    x = qkv.float().requires_grad_(True)
    ref, _ = _attn_ref(x, BT, N, H)
    ref.backward(do.float())
    tmp = os.path.join(tempfile.mkdtemp(), "libaim_two.so")
    shutil.copy(L.library_path(), tmp)
    monkeypatch.setenv("AIM_ATTN_BWD_PIPE", "0")
    lib2 = ctypes.CDLL(tmp)
    dq = torch.full((BT * N, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
    delta = torch.zeros((BT, H, N), device=DEV)
    lib2.aim_attn_bwd.argtypes = L.SIGNATURES["aim_attn_bwd"]
    rc = lib2.aim_attn_bwd(qkv.data_ptr(), out.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), BT, N, H,
                           torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    g = x.grad
    close(dq, g, 2e-2 * g.abs().max().item(), 3e-2, "two-kernel attn dqkv")
    assert float(delta.abs().max()) > 0.0           # this form writes delta = rowsum(dO o O) to the scratch


@pytest.mark.parametrize("BT,N,H,grid", [(3, 197, 12, 5), (2, 65, 1, 1), (2, 224, 2, 3), (5, 130, 3, 4), (4, 96, 2, 8), (30, 197, 12, 0),
                                         (3, 129, 2, 2), (2, 144, 1, 1), (3, 193, 1, 2), (2, 208, 2, 3), (2, 145, 1, 1), (2, 209, 2, 2)])
def test_attn_bwd_pipelined(BT, N, H, grid, monkeypatch):
    """The pipelined fused backward (the default for 65 <= N <= 224: persistent workgroups, loads one to two query blocks
    ahead across (frame, head) items) against plain PyTorch autograd.  AIM_ATTN_PIPE_GRID caps the grid so that a workgroup
    walks several items (item switches, K-image double buffer, deferred dK / dV stores) even at test sizes; run twice for
    bit-equality.  N = 64 j + 1 .. 64 j + 16 (j >= 2: 129..144, 193..208, so 197) takes the extra-tile form of the kernel."""
    _pipelined_case(BT, N, H, grid, monkeypatch)


@pytest.mark.parametrize("BT,N,H,grid", [(3, 197, 12, 5), (5, 130, 3, 4)])
def test_attn_bwd_pipelined_without_extra_tile(BT, N, H, grid, monkeypatch):
    """The same shapes with AIM_ATTN_PIPE_XT=0: the left-over queries get a tick of their own (the general form)."""
    monkeypatch.setenv("AIM_ATTN_PIPE_XT", "0")
    _pipelined_case(BT, N, H, grid, monkeypatch)


def _pipelined_case(BT, N, H, grid, monkeypatch):
    import ctypes, shutil, tempfile, os
    from aim_amd import lib as L
    ops = _ops()
    D = H * 64
    qkv = rnd((BT * N, 3 * D), 60, 1.0, torch.bfloat16)
    out = torch.empty((BT * N, D), dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros((BT, H, N), device=DEV)
    ops.attn_fwd(qkv, out, lse, BT, N, H)
    do = rnd((BT * N, D), 61, 1.0, torch.bfloat16)
    x = qkv.float().requires_grad_(True)
    ref, _ = _attn_ref(x, BT, N, H)
    ref.backward(do.float())
    tmp = os.path.join(tempfile.mkdtemp(), "libaim_pipe.so")
    shutil.copy(L.library_path(), tmp)
    if grid:
        monkeypatch.setenv("AIM_ATTN_PIPE_GRID", str(grid))
    lib2 = ctypes.CDLL(tmp)
    lib2.aim_attn_bwd.argtypes = L.SIGNATURES["aim_attn_bwd"]
    res = []
    for _ in range(2):
        dq = torch.full((BT * N, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
        delta = torch.zeros((BT, H, N), device=DEV)
        rc = lib2.aim_attn_bwd(qkv.data_ptr(), out.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), BT, N, H,
                               torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        res.append(dq)
    g = x.grad
    close(res[0], g, 2e-2 * g.abs().max().item(), 3e-2, "pipelined attn dqkv")
    assert torch.equal(res[0], res[1])


def test_attn_large_logits():
    """max-subtraction: logits of magnitude ~60 must not overflow."""
    ops = _ops()
    BT, N, H = 1, 40, 1
    qkv = rnd((BT * N, 192), 42, 3.0, torch.bfloat16)
    out = torch.zeros((BT * N, 64), dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros((BT, H, N), device=DEV)
    ops.attn_fwd(qkv, out, lse, BT, N, H)
    ref, lse_ref = _attn_ref(qkv, BT, N, H)
    close(lse, lse_ref, 5e-3, 1e-4, "attn lse large")
    close(out, ref, 3e-2, 3e-2, "attn out large")


@pytest.mark.parametrize("B,T,N,H", [(2, 2, 5, 2), (3, 8, 7, 12), (1, 32, 3, 1)])
def test_cls_attn(B, T, N, H):
    ops = _ops()
    D = H * 64
    BT = B * T
    qkv = rnd((BT * N, 3 * D), 43, 1.0, torch.bfloat16)
    out = torch.zeros((BT, D), dtype=torch.bfloat16, device=DEV)
    probs = torch.zeros((B, H, T, T), device=DEV)
    ops.cls_attn_fwd(qkv, out, probs, B, T, N, H)
    x = qkv.float().requires_grad_(True)
    c = x.reshape(BT, N, 3 * D)[:, 0]                                  # class rows
    q, k, v = [t.reshape(B, T, H, 64).permute(0, 2, 1, 3) for t in c.split(D, dim=1)]
    p = (q @ k.transpose(-1, -2) / 8.0).softmax(-1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(BT, D)
    close(probs, p, 1e-5, 1e-4, "cls probs")
    close(out, ref, 2e-3, 2 ** -7, "cls out")
    do = rnd((BT, D), 44, 1.0, torch.bfloat16)
    ref.backward(do.float())
    base = rnd((BT * N, 3 * D), 45, 1.0, torch.bfloat16)
    dqkv = base.clone()
    ops.cls_attn_bwd(qkv, probs, do, dqkv, B, T, N, H)
    close(dqkv, base.float() + x.grad, 2e-2, 2e-2, "cls dqkv (accumulated into class rows)")
    # compact mode: the class rows' share alone, [BT, 3D]; folded back by add_rows
    comp = torch.full((BT, 3 * D), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.cls_attn_bwd(qkv, probs, do, comp, B, T, N, H, compact=True)
    close(comp, x.grad.reshape(BT, N, 3 * D)[:, 0], 2e-2, 2e-2, "cls dqkv (compact)")
    tgt = base.clone()
    ops.add_rows(tgt, N * 3 * D, comp.float())
    close(tgt, base.float() + x.grad, 2e-2, 2e-2, "add_rows(class rows)")


def test_lambda():
    ops = _ops()
    BT, N, D = 4, 197, 768
    qkv = rnd((BT * N, 3 * D), 46, 0.35, torch.bfloat16)
    kx = rnd((BT, D), 47, 0.5, torch.bfloat16)
    nt = ops.expsum_tiles(N, N)
    part = torch.zeros((BT, nt, 2), device=DEV)
    q, k = qkv[:, :D], qkv[:, D:2 * D]
    ops.gemm(q, k, ops.EPI_EXPSUM, part, M=N, N=N, K=D, batch=BT, stride_a=N * 3 * D, stride_w=N * 3 * D,
             scale=0.125)
    lam, oml = torch.zeros(BT, device=DEV), torch.zeros(BT, device=DEV)
    ops.lambda_(qkv, kx, part, nt, lam, oml, BT, N, D, 0.125)
    # two-call form: the cross scores computed apart
    ss = torch.zeros((BT, N), device=DEV)
    ops.qk_cross(qkv, kx, ss, BT, N, D, 0.125)
    lam2, oml2 = torch.zeros(BT, device=DEV), torch.zeros(BT, device=DEV)
    ops.lambda_(qkv, kx, part, nt, lam2, oml2, BT, N, D, 0.125, ss=ss)
    close(lam2, lam, 1e-5, 1e-6, "lamda (two-call form)")
    # fused form: kx rides the ow GEMM as an extra key (16 slots per frame: 8 ow + 8 cw)
    part16 = torch.zeros((BT, 16, 2), device=DEV)
    ops.gemm(q, k, ops.EPI_EXPSUM, part16, M=N, N=N, K=D, batch=BT, stride_a=N * 3 * D, stride_w=N * 3 * D, scale=0.125,
             xrow=kx)
    lam3, oml3 = torch.zeros(BT, device=DEV), torch.zeros(BT, device=DEV)
    ops.lambda_partials(part16, lam3, oml3, BT)
    close(lam3, lam, 1e-4, 1e-6, "lamda (kx as an extra key of the ow GEMM)")
    close(oml3, 1.0 - lam, 1e-4, 1e-6, "1 - lamda")
    qf, kf = q.float().reshape(BT, N, D).double(), k.float().reshape(BT, N, D).double()
    ow = torch.exp(qf @ kf.transpose(1, 2) * 0.125).sum((1, 2))
    cw = torch.exp((qf @ kx.double().unsqueeze(-1)).squeeze(-1) * 0.125).sum(1)
    ref = (cw / (cw + ow)).float()
    close(lam, ref, 1e-7, 2e-4, "lambda")
    close(oml, 1 - ref, 1e-6, 1e-5, "1-lambda")


@pytest.mark.parametrize("BT,D", [(5, 1024), (3, 512)])
def test_lambda_border_form_n257(BT, D):
    """ViT-L/14 (N = 257): one full 256 x 256 EXPSUM tile per frame + aim_qk_border (the 257th row / column and the cross
    scores) against a float64 evaluation, and against the 3 x 3-tile form it replaces."""
    ops = _ops()
    N = 257
    qkv = rnd((BT * N, 3 * D), 146, 0.3, torch.bfloat16)
    kv = rnd((BT, 2 * D), 147, 0.4, torch.bfloat16)            # kx = the first D columns of the class chain's [k | v] rows
    q, k = qkv[:, :D], qkv[:, D:2 * D]
    part = torch.full((BT, 10, 2), float("nan"), device=DEV)
    ops.gemm(qkv, qkv[:, D:], ops.EPI_EXPSUM, part, M=N - 1, N=N - 1, K=D, batch=BT, stride_a=N * 3 * D, stride_w=N * 3 * D,
             scale=0.125, slot_stride=20)
    ss = torch.full((BT, N), float("nan"), device=DEV)
    ops.qk_border(qkv, kv, ss, part, 8, BT, N, D, 0.125)
    lam, oml = torch.zeros(BT, device=DEV), torch.zeros(BT, device=DEV)
    ops.lambda_(qkv, kv, part, 10, lam, oml, BT, N, D, 0.125, ss=ss)
    torch.cuda.synchronize()
    assert torch.isfinite(part).all() and torch.isfinite(ss).all()
    qf, kf = q.float().reshape(BT, N, D).double().cpu(), k.float().reshape(BT, N, D).double().cpu()
    kx = kv[:, :D].double().cpu()
    s = qf @ kf.transpose(1, 2) * 0.125
    c = (qf @ kx.unsqueeze(-1)).squeeze(-1) * 0.125
    close(ss, c.float().to(DEV), 1e-4, 1e-5, "cross scores")
    ow, cw = torch.exp(s).sum((1, 2)), torch.exp(c).sum(1)
    ref = (cw / (cw + ow)).float().to(DEV)
    close(lam, ref, 1e-7, 2e-4, "lamda (border form)")
    close(oml, 1 - ref, 1e-6, 1e-5, "1 - lamda (border form)")
    # the form it replaces: 3 x 3 tiles of 128 + qk_cross
    nt = ops.expsum_tiles(N, N)
    part9 = torch.zeros((BT, nt, 2), device=DEV)
    ops.gemm(q, k, ops.EPI_EXPSUM, part9, M=N, N=N, K=D, batch=BT, stride_a=N * 3 * D, stride_w=N * 3 * D, scale=0.125)
    lam9 = torch.zeros(BT, device=DEV)
    ops.lambda_(qkv, kv, part9, nt, lam9, None, BT, N, D, 0.125)
    close(lam, lam9, 1e-7, 1e-4, "border form vs tile form")


# ------------------------------------------------------------------ wgrad ----------------------
@pytest.mark.parametrize("M,Nw,Kw", [(64, 128, 128), (1000, 192, 768), (20, 32, 128), (4100, 768, 192), (512, 8, 16)])
def test_wgrad(M, Nw, Kw):
    ops = _ops()
    g, a = rnd((M, Nw), 50, 1.0, torch.bfloat16), rnd((M, Kw), 51, 1.0, torch.bfloat16)
    dw0, db0 = rnd((Nw, Kw), 52), rnd((Nw,), 53)
    dw, db = dw0.clone(), db0.clone()
    ops.wgrad(g, a, dw, db)
    ref = dw0 + g.float().T @ a.float()
    tol = 2e-5 * math.sqrt(M) * 4
    close(dw, ref, tol + 1e-4, 1e-4, "wgrad dW")
    close(db, db0 + g.float().sum(0), tol + 1e-4, 1e-4, "wgrad db")
    # bias gradient with a per-token row factor (DropPath-scaled D_fc2 bias, vit_clip.py:286), G and A row-strided views
    ntok = 197 if M > 400 else 5
    at = (torch.rand(ntok, generator=torch.Generator().manual_seed(54)) < 0.7).float().to(DEV) / 0.7
    wide_g = rnd((M, Nw + 64), 55, 1.0, torch.bfloat16)
    gv = wide_g[:, 32:32 + Nw]
    dw2, db2 = dw0.clone(), db0.clone()
    ops.wgrad(gv, a, dw2, db2, at=at, ntok=ntok)
    rs = at[torch.arange(M, device=DEV) % ntok]
    close(dw2, dw0 + gv.float().T @ a.float(), tol + 1e-4, 1e-4, "wgrad dW (strided G)")
    close(db2, db0 + (gv.float() * rs[:, None]).sum(0), tol + 1e-4, 1e-4, "wgrad db with row factors")
    # bitwise reproducible
    dw3, db3 = dw0.clone(), db0.clone()
    ops.wgrad(gv, a, dw3, db3, at=at, ntok=ntok)
    assert torch.equal(dw2, dw3) and torch.equal(db2, db3)


# ------------------------------------------------------------------ embed / misc ---------------
@pytest.mark.parametrize("p,res", [(16, 32), (14, 28)])
def test_patchify_and_embed(p, res):
    ops = _ops()
    B, T, D = 2, 3, 128
    G = res // p
    N = G * G + 1
    K = 3 * p * p
    Kp = (K + 63) // 64 * 64
    imgs = rnd((B, 3, T, res, res), 60)
    A = torch.full((B * T * G * G, Kp), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.patchify(imgs, A, B, T, res, res, p, Kp)
    ref = imgs.permute(0, 2, 1, 3, 4).reshape(B * T, 3, G, p, G, p).permute(0, 2, 4, 1, 3, 5).reshape(B * T * G * G, K)
    close(A[:, :K], ref, 1e-6, 2 ** -8, "patchify")
    assert (A[:, K:] == 0).all()
    # uint8 + GPUNormalize fused
    u8 = (torch.rand((B, 3, T, res, res), generator=torch.Generator().manual_seed(61)) * 255).to(torch.uint8).to(DEV)
    mean3 = torch.tensor([122.769, 116.74, 104.04], device=DEV)
    std3 = torch.tensor([68.493, 66.63, 70.321], device=DEV)
    ops.patchify(u8, A, B, T, res, res, p, Kp, mean3, std3)
    nrm = (u8.float() - mean3.view(1, 3, 1, 1, 1)) / std3.view(1, 3, 1, 1, 1)
    ref = nrm.permute(0, 2, 1, 3, 4).reshape(B * T, 3, G, p, G, p).permute(0, 2, 4, 1, 3, 5).reshape(B * T * G * G, K)
    close(A[:, :K], ref, 1e-6, 2 ** -8, "patchify uint8+normalize")
    # bf16 input, and an input whose base address is not 16-byte aligned (the 16-byte-load path must not be taken)
    ib = imgs.to(torch.bfloat16)
    ops.patchify(ib, A, B, T, res, res, p, Kp)
    refb = ib.float().permute(0, 2, 1, 3, 4).reshape(B * T, 3, G, p, G, p).permute(0, 2, 4, 1, 3, 5).reshape(B * T * G * G, K)
    close(A[:, :K], refb, 1e-6, 2 ** -8, "patchify bf16 input")
    flat = torch.zeros(imgs.numel() + 1, device=DEV)
    flat[1:] = imgs.reshape(-1)
    ops.patchify(flat[1:].view_as(imgs), A, B, T, res, res, p, Kp)
    ref = imgs.permute(0, 2, 1, 3, 4).reshape(B * T, 3, G, p, G, p).permute(0, 2, 4, 1, 3, 5).reshape(B * T * G * G, K)
    close(A[:, :K], ref, 1e-6, 2 ** -8, "patchify misaligned base")
    # embed + ln_pre, and its backward into temporal_embedding
    tok = rnd((B * T * G * G, D), 62, 1.0, torch.bfloat16)
    cls, pos = rnd((D,), 63), rnd((N, D), 64)
    tmp = rnd((T, D), 65).requires_grad_(True)
    gamma, beta = rnd((D,), 66) + 1, rnd((D,), 67)
    x = torch.zeros((B * T, N, D), device=DEV)
    mean, rstd = torch.zeros(B * T * N, device=DEV), torch.zeros(B * T * N, device=DEV)
    ops.embed_ln(tok, cls, pos, tmp.detach(), gamma, beta, x, mean, rstd, B, T, N, D)
    pre = torch.cat([cls.expand(B * T, 1, D), tok.float().reshape(B * T, G * G, D)], 1) + pos
    pre = (pre.reshape(B, T, N, D) + tmp.reshape(1, T, 1, D)).reshape(B * T, N, D)
    ref = torch.nn.functional.layer_norm(pre, (D,), gamma, beta, 1e-5)
    close(x, ref, 5e-6, 5e-6, "embed_ln")
    dx = rnd((B * T, N, D), 68)
    ref.backward(dx)
    dtmp = torch.zeros((T, D), device=DEV)
    ops.embed_bwd(dx, tok, cls, pos, tmp.detach(), gamma, mean, rstd, dtmp, B, T, N, D)
    close(dtmp, tmp.grad, 2e-4, 1e-4, "embed_bwd dtemporal")



def test_cast_table():
    """aim_cast_multi: row casts (4-wide path and the scalar one), 32x32-tile transposes (and the scalar one), fp32 bias copies
    into strided destinations, all in one launch."""
    ops = _ops()
    ents = []
    exp = []
    for i, (R, C, tr) in enumerate([(768, 192, False), (768, 192, True), (192, 768, True), (100, 36, True), (100, 36, False),
                                    (33, 7, False), (1, 192, 2), (1, 3264, 2)]):
        src = rnd((R, C), 200 + i)
        if tr == 2:
            big = torch.full((R, C + 64), float("nan"), device=DEV)
            dst = big[:, 32:32 + C]
            exp.append((dst, src.clone()))
        elif tr:
            big = torch.full((C, R + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
            dst = big[:, :R]
            exp.append((dst, src.t().to(torch.bfloat16)))
        else:
            big = torch.full((R, C + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
            dst = big[:, :C]
            exp.append((dst, src.to(torch.bfloat16)))
        ents.append((src, dst, tr))
    tab = ops.CastTable(ents, DEV)
    tab.run()
    torch.cuda.synchronize()
    for k, (got, want) in enumerate(exp):
        assert torch.equal(got, want), f"cast table entry {k}"

def test_misc_reductions_and_casts():
    ops = _ops()
    frames, ntok, D = 5, 7, 192
    x, w = rnd((frames * ntok, D), 70), rnd((ntok,), 71)
    out = torch.zeros((frames, D), device=DEV)
    ops.frame_sum(x, w, out, frames, ntok, D)
    close(out, (x.reshape(frames, ntok, D) * w[None, :, None]).sum(1), 1e-5, 1e-5, "frame_sum")
    for fr, nt, d in ((3, 197, 768), (2, 257, 1024), (4, 33, 64)):       # bf16 input, token counts that are not multiples of 4 / 32
        xb, wb = rnd((fr * nt, d), 75, 1.0, torch.bfloat16), rnd((nt,), 76)
        ob = torch.zeros((fr, d), device=DEV)
        ops.frame_sum(xb, wb, ob, fr, nt, d)
        close(ob, (xb.float().reshape(fr, nt, d) * wb[None, :, None]).sum(1), 1e-4, 1e-5, "frame_sum bf16")
        ops.frame_sum(xb, None, ob, fr, nt, d)
        close(ob, xb.float().reshape(fr, nt, d).sum(1), 1e-4, 1e-5, "frame_sum no weights")
    X = rnd((frames * ntok, D), 72, 1.0, torch.bfloat16)
    af, at = rnd((frames,), 73), rnd((ntok,), 74)
    cs = torch.zeros(D, device=DEV)
    ops.colsum(X, cs, af=af, at=at, ntok=ntok)
    rs = (af[:, None] * at[None, :]).reshape(-1, 1)
    close(cs, (rs * X.float()).sum(0), 1e-4, 1e-4, "colsum")
    src = rnd((50, 72), 75)
    d1 = torch.zeros((50, 72), dtype=torch.bfloat16, device=DEV)
    d2 = torch.zeros((72, 50), dtype=torch.bfloat16, device=DEV)
    ops.cast_bf16(src, d1)
    ops.cast_bf16(src, d2, transpose=True)
    assert torch.equal(d1, src.to(torch.bfloat16)) and torch.equal(d2, src.T.contiguous().to(torch.bfloat16))
    s = rnd((50,), 76)
    y = torch.zeros((50, 72), dtype=torch.bfloat16, device=DEV)
    ops.scale_rows(src, s, y=y)
    close(y, src * s[:, None], 1e-6, 2 ** -8, "scale_rows")


@pytest.mark.parametrize("M,C", [(5000, 768), (300, 192), (64, 32), (100, 20)])
def test_colsum_shapes(M, C):
    ops = _ops()
    X = rnd((M, C), 77, 1.0, torch.bfloat16)
    out = rnd((C,), 78)
    ref = out + X.float().sum(0)
    ops.colsum(X, out)
    close(out, ref, 2e-3, 1e-4, "colsum")


def test_flat_adamw_matches_torch():
    from aim_amd.dist import FlatAdamW
    torch.manual_seed(0)
    shapes = [(192, 768), (192,), (768, 192), (768,), (1, 8, 768), (7,)]
    ps = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ref = torch.optim.AdamW([dict(params=qs[:4], weight_decay=0.05), dict(params=qs[4:], weight_decay=0.0)],
                            lr=3e-4, betas=(0.9, 0.999))
    opt = FlatAdamW([dict(params=ps[:4], weight_decay=0.05), dict(params=ps[4:], weight_decay=0.0)], lr=3e-4,
                    betas=(0.9, 0.999), eps=1e-8)
    for it in range(3):
        opt.zero_grad(); ref.zero_grad()
        for p, q in zip(ps, qs):
            g = torch.randn_like(p)
            p.grad.add_(g)                    # autograd accumulates into the flat view the same way
            q.grad = g.clone()
        opt.step(); ref.step()
        for p, q in zip(ps, qs):
            close(p.detach(), q.detach(), 1e-6, 1e-5, f"adamw step {it}")
    assert all(p.data_ptr() >= opt.flat_p.data_ptr() for p in ps)


def test_errors_are_loud():
    ops = _ops()
    a = torch.zeros((8, 12), dtype=torch.bfloat16, device=DEV)   # K=12 not a multiple of 8
    w = torch.zeros((8, 12), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        ops.gemm(a, w, ops.EPI_BF16, torch.zeros((8, 8), dtype=torch.bfloat16, device=DEV))
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.gemm(a.cpu(), w, ops.EPI_BF16, torch.zeros((8, 8), dtype=torch.bfloat16, device=DEV))


# ---- classification tail: I3DHead + CrossEntropyLoss + top-k on HIP kernels ------------------------
def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("B,T,D,C,drop", [(64, 8, 768, 400, True), (3, 4, 128, 7, False), (1, 2, 64, 5, True)])
def test_head_and_ce_topk_kernels(B, T, D, C, drop):
    """aim_head_fwd/bwd + aim_ce_topk against the plain PyTorch fp32 ops of the reference's tail
    (i3d_head.py:53-73, cross_entropy_loss.py:78, accuracy.py:90-109)."""
    import numpy as np
    import torch.nn.functional as F
    from aim_amd import ops
    from aim_amd.recognizer import top_k_accuracy
    g = torch.Generator().manual_seed(B * 1000 + C)
    feat = torch.randn((B, T, D), generator=g).to(DEV)
    W = (torch.randn((C, D), generator=g) * 0.05).to(DEV).requires_grad_(True)
    b = (torch.randn((C,), generator=g) * 0.1).to(DEV).requires_grad_(True)
    label = torch.randint(0, C, (B,), generator=g).to(DEV)
    mask = (torch.empty((B, D)).bernoulli_(0.5, generator=g) / 0.5).to(DEV) if drop else None
    # reference ops
    fr = feat.clone().requires_grad_(True)
    pooled_ref = fr.mean(1) * (mask if drop else 1.0)
    score_ref = F.linear(pooled_ref, W, b)
    loss_ref = F.cross_entropy(score_ref, label)
    gf, gW, gb = torch.autograd.grad(loss_ref, [fr, W, b])
    # kernels
    pooled, score = ops.head_fwd(feat, mask, W.detach(), b.detach())
    out3, dscore = ops.ce_topk(score, label)
    dW, db, dfeat = ops.head_bwd(dscore, pooled, mask, W.detach(), T)
    torch.cuda.synchronize()
    assert _rel(score, score_ref) < 1e-5 and _rel(pooled, pooled_ref) < 1e-6
    assert abs(float(out3[0]) - float(loss_ref)) < 1e-5 * max(1.0, abs(float(loss_ref)))
    t1, t5 = top_k_accuracy(score_ref.detach().cpu().numpy(), label.cpu().numpy(), (1, 5))
    assert abs(float(out3[1]) - t1) < 1e-6 and abs(float(out3[2]) - t5) < 1e-6
    assert _rel(dW, gW) < 1e-5 and _rel(db, gb) < 1e-5 and _rel(dfeat, gf) < 1e-5


def test_head_modules_use_the_kernels_and_match_autograd():
    """I3DHead.forward / .loss on GPU tensors: same numbers as the eager path on a CPU copy, gradients included, and
    exact top-k tie handling (numpy argsort order) with tied scores."""
    import aim_amd
    torch.manual_seed(0)
    head = aim_amd.I3DHead(num_classes=11, in_channels=96, dropout_ratio=0.0)
    head.init_weights()
    ref = aim_amd.I3DHead(num_classes=11, in_channels=96, dropout_ratio=0.0)
    ref.load_state_dict(head.state_dict())
    head = head.to(DEV).train()
    x = torch.randn(6, 96, 4, 1, 1)
    lab = torch.tensor([0, 3, 3, 10, 7, 1])
    xs = x.to(DEV).requires_grad_(True)
    sc = head(xs)
    losses = head.loss(sc, lab.to(DEV))
    losses["loss_cls"].backward()
    xr = x.clone().requires_grad_(True)
    scr = ref(xr)
    lr = ref.loss(scr, lab)
    lr["loss_cls"].backward()
    assert set(losses) == {"top1_acc", "top5_acc", "loss_cls"}
    assert _rel(sc, scr) < 1e-5 and abs(float(losses["loss_cls"]) - float(lr["loss_cls"])) < 1e-5
    assert float(losses["top1_acc"]) == float(lr["top1_acc"]) and float(losses["top5_acc"]) == float(lr["top5_acc"])
    assert _rel(xs.grad, xr.grad) < 1e-5
    assert _rel(head.fc_cls.weight.grad, ref.fc_cls.weight.grad) < 1e-5 and _rel(head.fc_cls.bias.grad, ref.fc_cls.bias.grad) < 1e-5
    # ties: all-equal scores -> the label is in the top k iff fewer than k LARGER indices exist (numpy argsort order)
    from aim_amd import ops
    from aim_amd.recognizer import top_k_accuracy
    s = torch.zeros((4, 9), device=DEV)
    labs = torch.tensor([8, 4, 3, 0], device=DEV)
    out3, _ = ops.ce_topk(s, labs, need_grad=False)
    t1, t5 = top_k_accuracy(s.cpu().numpy(), labs.cpu().numpy(), (1, 5))
    assert float(out3[1]) == t1 and float(out3[2]) == t5
