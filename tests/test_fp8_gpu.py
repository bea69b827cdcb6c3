"""fp8 inference path (BASELINE configs[4]: fp8 MFMA weights, multi-view inference) on a real MI355X.

There is no reference counterpart (the reference infers in apex-O1 fp16), so the bar is defined here (SURVEY section 7,
hard part 7) in three layers:
  * kernel level -- ``aim_gemm_fp8`` against a plain fp32 matmul of the SAME fp8-grid operands: products of e4m3
    values are exact in fp32, so only the summation order differs (<= 1e-5 relative), for every epilogue;
    fp8-writing kernels (LayerNorm, attention, ACT8) against a CPU cast of their bf16 / fp32 twin's output;
  * model level -- the fp8 forward against the oracle's fp8 rounding-point emulation (``emu_backbone(f8=True)``);
  * task level -- top-1 agreement with the bf16 path wherever the bf16 top-1 margin exceeds the measured fp8 noise.
Every measured number is appended to gpurun_out/parity_r02.jsonl.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vit_clip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(case, **vals):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_r02.jsonl"), "a") as f:
        f.write(json.dumps(dict(case=case, **{k: float(v) for k, v in vals.items()})) + "\n")
    print("PARITY", case, {k: f"{float(v):.3e}" for k, v in vals.items()})


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _rand8(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(shape, generator=g) * scale
    q = O.q8(f)
    return q, q.to(torch.float8_e4m3fn).to(DEV)


@pytest.mark.parametrize("M,N,K", [(1024, 256, 256), (1576, 2304, 768), (2056, 1024, 1024), (1100, 776, 3264)])
def test_gemm_fp8_epilogues(M, N, K):
    from aim_amd import ops
    af, a8 = _rand8((M, K), 1)
    wf, w8 = _rand8((N, K), 2, 0.5)
    g = torch.Generator().manual_seed(3)
    ws = (torch.rand(N, generator=g) * 0.02 + 0.001)
    bias = torch.randn(N, generator=g) * 0.1
    ref = (af.double() @ wf.double().T * ws.double()[None]).float()        # exact products, fp64 sum
    # BF16
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    ops.gemm_fp8(a8, w8, ws.to(DEV), ops.EPI_BF16, out, bias=bias.to(DEV))
    want = (ref + bias).to(torch.bfloat16)
    assert _rel(out, want) < 2e-3                                        # one bf16 rounding of the output
    assert (out.float().cpu() - want.float()).abs().max() <= 2 ** -7 * want.float().abs().max()
    # F32 with residual + per-frame factor + per-token vector
    ntok = 197 if M % 197 == 0 else 128
    if M % ntok:
        ntok = M            # one "frame"
    nfr = M // ntok
    resid = torch.randn((M, N), generator=g)
    afac = torch.rand(nfr, generator=g)
    vec = torch.randn((nfr, N), generator=g)
    bt = torch.rand(ntok, generator=g)
    out32 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    kw = dict(bias=bias.to(DEV), resid=resid.to(DEV), af=afac.to(DEV), vec=vec.to(DEV), bt=bt.to(DEV), ntok=ntok) if ntok >= 128 else \
        dict(bias=bias.to(DEV), resid=resid.to(DEV))
    ops.gemm_fp8(a8, w8, ws.to(DEV), ops.EPI_F32, out32, **kw)
    if ntok >= 128:
        rows = torch.arange(M)
        want32 = resid + afac[rows // ntok][:, None] * (ref + bias) + bt[rows % ntok][:, None] * vec[rows // ntok]
    else:
        want32 = resid + ref + bias
    assert _rel(out32, want32) < 1e-5, _rel(out32, want32)
    # RES16: the same sum on a bf16 residual stream, rounded to bf16 once
    resid16 = resid.to(torch.bfloat16)
    out16 = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    kw16 = dict(kw, resid=resid16.to(DEV))
    ops.gemm_fp8(a8, w8, ws.to(DEV), ops.EPI_RES16, out16, **kw16)
    want16 = (want32 - resid + resid16.float())
    assert _rel(out16, want16.to(torch.bfloat16)) < 2e-3
    assert (out16.float().cpu() - want16).abs().max() <= 2 ** -7 * want16.abs().max()          # one bf16 rounding of the sum
    # ACT8 with a column split (QuickGELU | at * GELU) -> fp8 bytes
    ns = (N // 2) // 8 * 8
    at = torch.rand(ntok, generator=g) if ntok >= 128 else None
    o8 = torch.empty((M, N), dtype=torch.float8_e4m3fn, device=DEV)
    kw = dict(at=at.to(DEV), ntok=ntok) if at is not None else {}
    ops.gemm_fp8(a8, w8, ws.to(DEV), ops.EPI_ACT8, o8, bias=bias.to(DEV), act=ops.ACT_QGELU, n_split=ns, act2=ops.ACT_GELU, **kw)
    pre = ref + bias
    rs = at[torch.arange(M) % ntok][:, None] if at is not None else 1.0
    want8 = torch.cat([pre[:, :ns] * torch.sigmoid(1.702 * pre[:, :ns]), rs * torch.nn.functional.gelu(pre[:, ns:])], 1)
    got = o8.float().cpu()
    wq = O.q8(want8)
    # same e4m3 grid point except where fp32 noise lands on a rounding boundary (one fp8 ulp = 2^-3 relative)
    frac_exact = (got == wq).float().mean().item()
    assert frac_exact > 0.995, frac_exact
    assert _rel(got, wq) < 5e-3


def test_gemm_fp8_rate_vs_bf16():
    """The block-scaled MFMA form runs K = 128 per instruction: at the same shape the fp8 GEMM must beat the bf16 one
    (recorded, asserted only loosely: devices differ)."""
    from aim_amd import ops
    M, N, K = 65792, 3072, 1024                    # ViT-L/14, 8 views x 32 frames x 257 tokens
    a = torch.randn((M, K), device=DEV)
    w = torch.randn((N, K), device=DEV) * 0.03
    a16, w16 = a.to(torch.bfloat16), w.to(torch.bfloat16)
    a8 = a.clamp(-448, 448).to(torch.float8_e4m3fn)
    w8, ws = ops.quantize_fp8_rows(w)
    o16 = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    o8 = torch.empty_like(o16)

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10

    t16 = timeit(lambda: ops.gemm(a16, w16, ops.EPI_BF16, o16))
    t8 = timeit(lambda: ops.gemm_fp8(a8, w8, ws, ops.EPI_BF16, o8))
    fl = 2.0 * M * N * K
    _record("gemm_fp8_vs_bf16_L14", bf16_ms=t16, fp8_ms=t8, bf16_tflops=fl / t16 / 1e9, fp8_tflops=fl / t8 / 1e9,
            rel_err_fp8_vs_bf16=_rel(o8, o16))
    assert t8 < 0.9 * t16, (t8, t16)


def test_fp8_producers_match_their_twins():
    """LayerNorm / attention with fp8 output == saturating e4m3 cast of the bf16 kernels' fp32 / bf16 results."""
    from aim_amd import ops
    rows, D = 1500, 1024
    x = torch.randn((rows, D), device=DEV) * 3
    g, b = torch.randn(D, device=DEV), torch.randn(D, device=DEV)
    y32 = torch.empty((rows, D), device=DEV)
    ops.layernorm_fwd(x, g, b, rows, D, D, y_f32=y32)
    y8 = torch.empty((rows, D), dtype=torch.float8_e4m3fn, device=DEV)
    ops.layernorm_fwd_fp8(x, g, b, rows, D, D, y8)
    assert torch.equal(y8.float().cpu(), O.q8(y32.cpu()))
    big = torch.full((8, D), 1e4, device=DEV); big[:, ::2] = -1e4      # saturation: |y| > 448 clamps, never NaN
    y8b = torch.empty((8, D), dtype=torch.float8_e4m3fn, device=DEV)
    ops.layernorm_fwd_fp8(big, torch.full((D,), 1e3, device=DEV), b, 8, D, D, y8b)
    assert torch.isfinite(y8b.float()).all() and y8b.float().abs().max() == 448
    BT, N, H = 6, 257, 16
    qkv = (torch.randn((BT * N, 3 * H * 64), device=DEV) * 0.7).to(torch.bfloat16)
    o16 = torch.empty((BT * N, H * 64), dtype=torch.bfloat16, device=DEV)
    lse = torch.empty((BT, H, N), device=DEV)
    ops.attn_fwd(qkv, o16, lse, BT, N, H)
    o8 = torch.empty((BT * N, H * 64), dtype=torch.float8_e4m3fn, device=DEV)
    ops.attn_fwd_fp8(qkv, o8, BT, N, H)
    got, want = o8.float().cpu(), O.q8(o16.float().cpu())
    # o8 is cast from the fp32 accumulators, o16 went through bf16 first: grid points may differ by one fp8 ulp
    assert (got == want).float().mean() > 0.97 and _rel(got, want) < 2e-2


def test_layernorm_bf16_rows_in():
    """aim_layernorm_fwd_x16 (the fp8 path's bf16 residual stream) against the fp32-input kernel on the same bf16 values."""
    from aim_amd import ops
    rows, D, N = 600, 1024, 3
    x16 = (torch.randn((rows * N, D), generator=torch.Generator().manual_seed(9)) * 2 + 0.3).to(torch.bfloat16).to(DEV)
    g_, b_ = torch.rand(D, device=DEV) + 0.5, torch.randn(D, device=DEV) * 0.1
    ya, yb = torch.empty((rows, D), device=DEV), torch.empty((rows, D), device=DEV)
    ops.layernorm_fwd_x16(x16, g_, b_, rows, D, N * D, y_f32=ya)                 # strided rows (class tokens)
    ops.layernorm_fwd(x16.float(), g_, b_, rows, D, N * D, y_f32=yb)
    assert torch.equal(ya, yb)
    y8a = torch.empty((rows * N, D), dtype=torch.uint8, device=DEV)
    y8b = torch.empty_like(y8a)
    y16 = torch.empty((rows * N, D), dtype=torch.bfloat16, device=DEV)
    ops.layernorm_fwd_x16(x16, g_, b_, rows * N, D, D, y8=y8a, y_bf16=y16)
    ops.layernorm_fwd_fp8(x16.float(), g_, b_, rows * N, D, D, y8b)
    assert torch.equal(y8a, y8b)


def _model(res, T, patch, D, L, H, seed):
    import aim_amd
    m = aim_amd.ViT_CLIP(res, T, patch, D, L, H, 0.0)
    m.init_weights()
    st = O.synth_state_dict(O.backbone_param_shapes(res, T, patch, D, L), seed=seed)
    m.load_state_dict(st, strict=True)
    return m.to(DEV).eval(), st


@pytest.mark.parametrize("arch", ["B16_L2", "L14_L2"])
def test_fp8_forward_matches_rounding_point_oracle(arch):
    res, T, patch, D, L, H, B = (224, 2, 16, 768, 2, 12, 4) if arch == "B16_L2" else (224, 4, 14, 1024, 2, 16, 1)
    m, st = _model(res, T, patch, D, L, H, 71)
    imgs = torch.randn((B, 3, T, res, res), generator=torch.Generator().manual_seed(72))
    with torch.no_grad():
        y16 = m(imgs.to(DEV))
        m.set_inference_precision('fp8')
        y8 = m(imgs.to(DEV))
        e8 = O.emu_backbone(imgs, st, H, rnd=O.BF16, f8=True)
        e16 = O.emu_backbone(imgs, st, H, rnd=O.BF16)
    vals = dict(fp8_vs_emu8=_rel(y8, e8), fp8_vs_bf16=_rel(y8, y16), emu8_vs_emu16=_rel(e8, e16), bf16_vs_emu16=_rel(y16, e16))
    _record("fp8_forward_" + arch, **vals)
    assert not torch.equal(y8, y16)                      # the fp8 kernels really ran
    # Noise floor: two implementations of a chain of rounded stages that differ only in fp32
    # summation order drift apart until flip probability (delta / ulp) and flip size (ulp) balance: delta ~ 0.26 * ulp.
    # bf16 (ulp 2^-8): 1e-3, as measured on the bf16 path; fp8 e4m3 (ulp 2^-4): 1.6e-2 -- measured 2.4e-2 here, about
    # half of the fp8-vs-bf16 distance itself (4.7e-2).  The kernels' arithmetic is pinned exactly at kernel level
    # (test_gemm_fp8_epilogues, test_fp8_producers_match_their_twins); this bound only guards the wiring.
    assert vals["fp8_vs_emu8"] < 0.75 * vals["emu8_vs_emu16"], vals
    assert abs(vals["fp8_vs_bf16"] / vals["emu8_vs_emu16"] - 1.0) < 0.5, vals
    # grad-enabled forwards (training) never take the fp8 path
    y_tr = m(imgs.to(DEV))
    assert torch.equal(y_tr.detach(), y16)


def test_fp8_multiview_inference_agrees_with_bf16():
    """configs[4] shape per view at reduced depth/frames: Recognizer3D._do_test with 3 views per sample,
    max_testing_views chunking and clip averaging.  Tolerance (defined here; the reference has no fp8 mode), in logit
    space with ONE global noise level (the RMS fp8-vs-bf16 logit deviation over the whole batch, not a sample's own
    deviation): every pair of classes that the bf16 path separates by more than 5 * sqrt(2) * noise must be ranked
    the same way by the fp8 path -- in particular the class index is bit-exact wherever the top-1 margin exceeds that.
    (A random-init backbone gives nearly input-independent features, so top-1 margins of synthetic samples are one
    unlucky draw of the head's order statistics; the pairwise form uses all 400 classes of every sample.)"""
    import aim_amd
    T, L, C, S = 8, 4, 400, 24
    cfg = dict(type='Recognizer3D',
               backbone=dict(type='ViT_CLIP', input_resolution=224, patch_size=14, num_frames=T, width=1024, layers=L,
                             heads=16, drop_path_rate=0.0, adapter_scale=0.5, pretrained=None),
               cls_head=dict(type='I3DHead', in_channels=1024, num_classes=C, spatial_type='avg', dropout_ratio=0.5, init_std=0.05),
               test_cfg=dict(average_clips='score', max_testing_views=2))
    torch.manual_seed(5)
    model = aim_amd.build_model(cfg)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "D_fc2" in n:
                p.normal_(0, 0.02)
    model = model.to(DEV).eval()
    gen = torch.Generator().manual_seed(6)
    samples = [torch.randn((1, 3, 3, T, 224, 224), generator=gen) for _ in range(S)]       # [1, V=3, 3, T, H, W]
    with torch.no_grad():
        s16 = np.concatenate([model(s.to(DEV), return_loss=False) for s in samples])
        model.backbone.set_inference_precision('fp8')
        s8 = np.concatenate([model(s.to(DEV), return_loss=False) for s in samples])
        model.test_cfg['average_clips'] = 'prob'
        p8 = model(samples[0].to(DEV), return_loss=False)
    assert s16.shape == (S, C) and p8.shape == (1, C) and abs(p8.sum() - 1) < 1e-4
    noise = float(np.sqrt(np.mean((s8 - s16) ** 2)))
    tol = 5.0 * np.sqrt(2.0) * noise
    d16 = s16[:, :, None] - s16[:, None, :]
    d8 = s8[:, :, None] - s8[:, None, :]
    sep = d16 > tol
    top2 = np.sort(s16, axis=1)[:, -2:]
    margin_ok = (top2[:, 1] - top2[:, 0]) > tol
    agree_all = float((s8.argmax(1) == s16.argmax(1)).mean())
    _record("fp8_multiview_ranking", logit_rms_noise=noise, logit_std=float(s16.std()), tol=tol, pairs_separated=float(sep.mean() * 2),
            pairs_misranked=int((d8[sep] <= 0).sum()), top1_agree_all=agree_all, top1_margin_ok=int(margin_ok.sum()), n_samples=S,
            rel=_rel(torch.from_numpy(s8), torch.from_numpy(s16)))
    assert sep.mean() * 2 > 0.5                     # the criterion covers most class pairs
    assert (d8[sep] > 0).all()
    assert (s8.argmax(1)[margin_ok] == s16.argmax(1)[margin_ok]).all()


def test_cfg4_full_shape_three_views():
    """BASELINE configs[4] at its FULL per-sample shape: ViT-L/14 + AIM, 24 layers, 32 frames 224^2, one sample x 3 views
    through ``Recognizer3D._do_test`` (``max_testing_views=4`` as configs/recognition/vit/vitclip_large_k400.py:8,
    ``average_clips='prob'``), fp8 and bf16 (recognizer3d.py:31-85).  A CPU oracle of this size takes minutes per view,
    so the comparator at full size is the GPU's own reference-precision path (``set_precision('fp32')``: f32-MFMA kernels
    held to the REAL reference's fixtures at 1e-5 by tests/test_fp32_gpu.py), plus the size-independent properties:
    determinism, finiteness, probabilities that sum to one, no per-block context kept."""
    import aim_amd
    T, L, C, V = 32, 24, 400, 3
    cfg = dict(type='Recognizer3D',
               backbone=dict(type='ViT_CLIP', input_resolution=224, patch_size=14, num_frames=T, width=1024, layers=L,
                             heads=16, drop_path_rate=0.2, adapter_scale=0.5, pretrained=None),
               cls_head=dict(type='I3DHead', in_channels=1024, num_classes=C, spatial_type='avg', dropout_ratio=0.5, init_std=0.05),
               test_cfg=dict(average_clips='prob', max_testing_views=4))
    torch.manual_seed(17)
    model = aim_amd.build_model(cfg)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "D_fc2" in n or "temporal_embedding" in n:
                p.normal_(0, 0.02)
    model = model.to(DEV).eval()
    bb = model.backbone
    imgs = torch.randn((1, V, 3, T, 224, 224), generator=torch.Generator().manual_seed(18)).to(DEV)
    views = imgs[0]
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        p16 = model._do_test(imgs)
        p16b = model._do_test(imgs)
        f16 = bb(views).float()
        peak16 = torch.cuda.max_memory_allocated() - base
        bb.set_inference_precision('fp8')
        p8 = model._do_test(imgs)
        p8b = model._do_test(imgs)
        f8 = bb(views).float()
        bb.set_inference_precision('bf16').set_precision('fp32')
        f32 = bb(views).float()
        p32 = model._do_test(imgs)
        bb.set_precision('bf16')
    assert p16.shape == (1, C) and f16.shape == (V, 1024, T, 1, 1)
    for p in (p16, p8, p32):
        assert torch.isfinite(p).all() and abs(float(p.sum()) - 1.0) < 1e-4
    assert torch.equal(p16, p16b) and torch.equal(p8, p8b)                 # deterministic, both precisions
    assert not torch.equal(p8, p16)                                        # the fp8 kernels really ran
    vals = dict(bf16_vs_fp32=_rel(f16, f32), fp8_vs_fp32=_rel(f8, f32), fp8_vs_bf16=_rel(f8, f16),
                prob_bf16_vs_fp32=float((p16 - p32).abs().max()), prob_fp8_vs_fp32=float((p8 - p32).abs().max()),
                top1_bf16=float(p16.argmax(1) == p32.argmax(1)), top1_fp8=float(p8.argmax(1) == p32.argmax(1)),
                top2_margin_fp32=float(p32.topk(2).values.diff().abs()), peak_gib=peak16 / 2 ** 30)
    _record("cfg4_full_shape", **vals)
    # bf16 product vs reference-precision arithmetic through 24 blocks: the 24-layer noise floor of DESIGN.md section 5
    # (3.4e-3 against the bf16-rounding emulation) plus the emulation's own distance from fp32
    # (measured: bf16 5.0e-3, fp8 7.1e-2 from fp32; probabilities 4.8e-4 / 5.5e-3; both top-1 equal to fp32's)
    assert vals["bf16_vs_fp32"] < 1e-2, vals
    # fp8: e4m3 has 2^-4 relative resolution against bf16's 2^-9; same bound form as the 2-layer oracle test, from fp32
    assert vals["fp8_vs_fp32"] < 0.15 and vals["fp8_vs_bf16"] < 0.15, vals
    assert vals["prob_bf16_vs_fp32"] < 1e-3 and vals["prob_fp8_vs_fp32"] < 1.2e-2, vals
    assert peak16 < 24 * 2 ** 30, peak16          # no-grad forwards keep no per-block context (3 views x 8 224 rows x 24 blocks)
