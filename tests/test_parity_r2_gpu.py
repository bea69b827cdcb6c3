"""Round-2 parity cases on a real MI355X (VERDICT r1 "next round" item 1):

* DropPath ACTIVE (train mode) at block and backbone level, fed the masks the REAL reference drew
  (tests/golden/*_droppath.npz; zeros included, two different masks per block: vit_clip.py:112,275,286);
* measured errors at 2 / 12 / 24 layers against the same-rounding-point oracle ``emu_*(BF16)``, forward and
  per-tensor gradients -- every number is appended to gpurun_out/parity_r02.jsonl and quoted in DESIGN.md section 5;
* BASELINE configs[1] at its real size (64 clips) and the configs[3] per-GPU shape (ViT-L/14, 16 frames, 32 clips)
  through size-independent properties;
* the stream schedule as an explicit race check: side streams on vs off must be bit-identical.
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
        f.write(json.dumps(dict(case=case, **{k: (float(v) if not isinstance(v, (dict, str)) else v) for k, v in vals.items()})) + "\n")
    print("PARITY", case, {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in vals.items()})


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float32)


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _max(a, b):
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()


def _model(res, T, patch, D, L, H, seed, drop=0.0):
    import aim_amd
    m = aim_amd.ViT_CLIP(res, T, patch, D, L, H, drop)
    m.init_weights()
    st = O.synth_state_dict(O.backbone_param_shapes(res, T, patch, D, L), seed=seed)
    m.load_state_dict(st, strict=True)
    return m.to(DEV).eval(), st


def _run_block(m, layer, x_nbd, g_nbd, B, T, N, H, m1, m2, scale=0.5):
    """One HIP block forward + backward on a reference-layout input [N, BT, D] with explicit DropPath factors."""
    from aim_amd import backbone as bb
    D = x_nbd.shape[-1]
    fz = m._frozen_operands()["blocks"][layer]
    blk = m.transformer.resblocks[layer]
    adp = {a: bb._AdapterW(getattr(blk, a).D_fc1.weight, getattr(blk, a).D_fc1.bias, getattr(blk, a).D_fc2.weight,
                           getattr(blk, a).D_fc2.bias) for a in bb._ADAPTERS if a != "MLP_Adapter"}
    ma = blk.MLP_Adapter
    fz.stage_mlp_adapter(ma.D_fc1.weight, ma.D_fc1.bias, ma.D_fc2.weight, ma.D_fc2.bias)
    x = x_nbd.permute(1, 0, 2).contiguous().reshape(B * T * N, D).to(DEV)
    dms1, dms2 = (m1 * scale).to(DEV).contiguous(), (m2 * scale).to(DEV).contiguous()
    y, c = bb._block_forward(x, fz, adp, B, T, N, H, dms1, dms2, True)

    def _param(a, leaf):
        mod, attr = leaf.split(".")
        return getattr(getattr(getattr(blk, a), mod), attr)
    grads = {a: {leaf: torch.zeros_like(_param(a, leaf), dtype=torch.float32) for leaf in bb._ADAPTER_LEAVES}
             for a in bb._ADAPTERS}
    g = g_nbd.permute(1, 0, 2).contiguous().reshape(B * T * N, D).to(DEV)
    dx = bb._block_backward(g.to(torch.bfloat16), c, fz, adp, grads, B, T, N, H)
    torch.cuda.synchronize()
    to_nbd = lambda t: t.float().reshape(B * T, N, D).permute(1, 0, 2).cpu()
    return to_nbd(y), to_nbd(dx), {f"{a}.{leaf}": grads[a][leaf].cpu() for a in grads for leaf in grads[a]}, c


def _emu_block_grads(st, layer, x_nbd, g_nbd, H, T, m1, m2):
    names = [f"{a}.{leaf}" for a in ("MLP_Adapter", "S_Adapter", "T_Adapter")
             for leaf in ("D_fc1.weight", "D_fc1.bias", "D_fc2.weight", "D_fc2.bias")]
    full = [f"transformer.resblocks.{layer}.{n}" for n in names]
    for n in full:
        st[n] = st[n].detach().requires_grad_(True)
    xe = x_nbd.permute(1, 0, 2).contiguous().requires_grad_(True)
    # the product carries the upstream gradient in bf16
    ge = g_nbd.permute(1, 0, 2).to(torch.bfloat16).float()
    ye = O.emu_block(xe, st, layer, H, T, 0.5, O.BF16, drop_mask=(m1, m2))
    gr = torch.autograd.grad(ye, [xe] + [st[n] for n in full], ge)
    return ye.detach().permute(1, 0, 2), gr[0].permute(1, 0, 2), dict(zip(names, gr[1:]))


# Per-tensor gradient bounds (relative L2) vs the same-rounding-point oracle's autograd.  The product rounds the
# residual-stream gradient to bf16 once per LayerNorm backward (twice per block) and every dgrad operand to bf16
# (2^-9 relative each); the oracle's straight-through autograd keeps fp32.  One block: measured 3e-3 .. 7e-3, bound 1e-2.
# Through L blocks the roundings add in quadrature (~ sqrt(2 L) * 2^-9): measured median over all trainable tensors
# 3.6e-3 / 7.8e-3 / 9.8e-3 at 2 / 12 / 24 layers; the worst tensor is always one on the class-token path of a LOW
# layer (T_/S_Adapter: its gradient is a sum over B*T rows only -- 2 to 4 rows in these one-clip cases -- so nothing
# averages the rounding out): 1.2e-2 / 3.2e-2 / 2.8e-2.  Bounds leave ~1.5x.
GRAD_EMU_BOUND = 1e-2                                    # single block
GRAD_EMU_DEPTH = {2: (2e-2, 8e-3), 12: (5e-2, 1.5e-2), 24: (5e-2, 1.5e-2)}      # layers -> (worst tensor, median)
GRAD_REF_BOUND = 2.5e-2      # vs the real reference's fp32 autograd: adds the forward's bf16 error


@pytest.mark.parametrize("T", [2, 4])
def test_block_droppath_tiny(golden_dir, T):
    z = _load(golden_dir, f"block_tiny_T{T}_droppath.npz")
    D, H, N, B, T_, seed = [int(v) for v in z["meta"]]
    m, st = _model(32, T, 16, D, 2, H, seed)
    y, dx, grads, c = _run_block(m, 1, z["x"], z["g"], B, T, N, H, z["m1"], z["m2"])
    ye, dxe, ge = _emu_block_grads(st, 1, z["x"], z["g"], H, T, z["m1"], z["m2"])
    e = dict(y_emu_rel=_rel(y, ye), y_emu_max=_max(y, ye), y_ref_rel=_rel(y, z["y"]), dx_emu=_rel(dx, dxe),
             dx_ref=_rel(dx, z["dx"]), grad_emu=max(_rel(grads[n], ge[n]) for n in ge),
             grad_ref=max(_rel(grads[n], z["grad." + n]) for n in ge))
    _record(f"block_droppath_tiny_T{T}", **e)
    assert e["y_emu_rel"] < 1e-3 and e["y_emu_max"] < 8e-3, e       # the 1e-3 (bf16) bar, <= 2 bf16 ulps at |x| ~ 3
    assert e["y_ref_rel"] < 8e-3, e
    assert e["dx_emu"] < GRAD_EMU_BOUND and e["dx_ref"] < GRAD_REF_BOUND, e
    assert e["grad_emu"] < GRAD_EMU_BOUND and e["grad_ref"] < GRAD_REF_BOUND, e
    # dropped token positions: the adapter terms vanish there, so a zero in m2 must show in the MLP_Adapter path --
    # compare against the SAME block without masks to be sure the masks were applied at all
    y0, _, _, _ = _run_block(m, 1, z["x"], z["g"], B, T, N, H, torch.ones(N), torch.ones(N))
    assert _max(y, y0) > 1e-2


def test_block_droppath_real_shape(golden_dir):
    """N = 197, D = 768: the persistent 256^2 GEMM's per-token epilogue factors (at / bt / vec / b2row), colsum(at=)
    and frame_sum(w=) with zero-valued and distinct masks."""
    z = _load(golden_dir, "block_real_T2_droppath.npz")
    D, H, N, B, T, seed = [int(v) for v in z["meta"]]
    m, st = _model(224, T, 16, D, 2, H, seed)
    x, g = _randn((N, B * T, D), seed + 1), _randn((N, B * T, D), seed + 2)
    y, dx, grads, c = _run_block(m, 1, x, g, B, T, N, H, z["m1"], z["m2"])
    ye, dxe, ge = _emu_block_grads(st, 1, x, g, H, T, z["m1"], z["m2"])

    def samp(t, key):        # relative L2 error on the fixture's sampled elements of the REAL reference's tensor
        got = t.reshape(-1)[z[key + ".idx"].long()]
        return _rel(got, z[key + ".val"])
    e = dict(y_emu_rel=_rel(y, ye), y_emu_max=_max(y, ye), y_ref_rel=samp(y, "y"), dx_emu=_rel(dx, dxe),
             dx_ref=samp(dx, "dx"), grad_emu=max(_rel(grads[n], ge[n]) for n in ge),
             grad_ref=max(samp(grads[n], "grad." + n) for n in ge),
             per_tensor_emu={n: _rel(grads[n], ge[n]) for n in ge})
    _record("block_droppath_real_T2", **e)
    # At K = 768 / 3072 two implementations that differ only in fp32 summation order disagree on ~1 % of the bf16
    # rounding decisions of every stored intermediate, and each flipped element (one bf16 ulp = 2^-8 relative) perturbs
    # the next stage: the chain settles at a relative L2 distance of ~1e-3 whatever the implementation (DESIGN.md
    # section 5, "noise floor").  Measured 1.09e-3 here (tiny shapes: 1e-7 .. 2e-4); the bound leaves 2x.
    assert e["y_emu_rel"] < 2e-3 and e["y_emu_max"] < 1.6e-2, e
    assert e["y_ref_rel"] < 8e-3, e
    assert e["dx_emu"] < GRAD_EMU_BOUND and e["dx_ref"] < GRAD_REF_BOUND, e
    assert e["grad_emu"] < GRAD_EMU_BOUND and e["grad_ref"] < GRAD_REF_BOUND, e


def test_backbone_droppath_tiny(golden_dir, monkeypatch):
    """Whole backbone in train mode with the reference's drawn masks injected into ``ViT_CLIP._drop_mask``."""
    import aim_amd
    z = _load(golden_dir, "backbone_tiny_T2_droppath.npz")
    D, H, L, B, T, seed = [int(v) for v in z["meta"]]
    m, st = _model(32, T, 16, D, L, H, seed, drop=0.5)
    m.train()
    mk = z["masks"]
    real = m._drop_masks
    calls = []

    def fake(N, training, dev):
        own = real(N, training, dev)                      # shape / layer-0 semantics come from the product
        assert training and tuple(own.shape) == (L, 2, N)
        assert torch.equal(own[0], torch.full((2, N), 0.5, device=dev))     # layer 0: rate 0 = Identity (vit_clip.py:297)
        vals = sorted(set(own[2].flatten().tolist()))
        assert vals[0] == 0.0 and abs(vals[-1] - 0.5 / 0.5) < 1e-6          # layer 2: rate 0.5 -> {0, scale / keep}
        out = own.clone()
        out[1, 0], out[1, 1], out[2, 0], out[2, 1] = [(mk[j] * 0.5).to(dev) for j in range(4)]
        calls.append(1)
        return out
    monkeypatch.setattr(m, "_drop_masks", fake)
    y = m(z["imgs"].to(DEV))
    y.backward(z["g"].to(DEV))
    assert len(calls) == 1
    names = O.trainable_names(st)
    for n in names:
        st[n] = st[n].detach().requires_grad_(True)
    masks = [None, (mk[0], mk[1]), (mk[2], mk[3])]
    ye = O.emu_backbone(z["imgs"], st, H, rnd=O.BF16, drop_masks=masks)
    ge = dict(zip(names, torch.autograd.grad(ye, [st[n] for n in names], z["g"])))
    got = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    e = dict(y_emu_rel=_rel(y, ye), y_emu_max=_max(y, ye), y_ref_rel=_rel(y, z["y"]),
             grad_emu=max(_rel(got[n], ge[n]) for n in names), grad_ref=max(_rel(got[n], z["grad." + n]) for n in names))
    _record("backbone_droppath_tiny_T2", **e)
    assert e["y_emu_rel"] < 3e-3 and e["y_emu_max"] < 1.2e-2 and e["y_ref_rel"] < 1.5e-2, e
    assert e["grad_emu"] < GRAD_EMU_DEPTH[2][0] and e["grad_ref"] < GRAD_REF_BOUND, e


# ---- errors by depth -------------------------------------------------------------------------------------------
def _depth_case(tag, res, T, patch, D, L, H, seed, B, with_grads, y_rel_bound, y_max_bound):
    m, st = _model(res, T, patch, D, L, H, seed)
    imgs = _randn((B, 3, T, res, res), seed + 1)
    g = _randn((B, D, T, 1, 1), seed + 2)
    names = O.trainable_names(st)
    if with_grads:
        y = m(imgs.to(DEV))
        y.backward(g.to(DEV))
        for n in names:
            st[n] = st[n].detach().requires_grad_(True)
        ye = O.emu_backbone(imgs, st, H, rnd=O.BF16)
        ge = dict(zip(names, torch.autograd.grad(ye, [st[n] for n in names], g)))
        got = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
        per = {n: _rel(got[n], ge[n]) for n in names}
        worst = max(per, key=per.get)
        e = dict(y_emu_rel=_rel(y, ye), y_emu_max=_max(y, ye), grad_emu_worst=per[worst], grad_emu_worst_name=worst,
                 grad_emu_median=float(np.median(list(per.values()))))
    else:
        with torch.no_grad():
            y = m(imgs.to(DEV))
            ye = O.emu_backbone(imgs, st, H, rnd=O.BF16)
        e = dict(y_emu_rel=_rel(y, ye), y_emu_max=_max(y, ye))
    _record(tag, layers=L, **e)
    assert e["y_emu_rel"] < y_rel_bound and e["y_emu_max"] < y_max_bound, e
    if with_grads:
        worst_b, med_b = GRAD_EMU_DEPTH[L]
        assert e["grad_emu_worst"] < worst_b and e["grad_emu_median"] < med_b, e
    return e


def test_errors_at_2_layers():
    _depth_case("depth2_tiny_T4", 32, 4, 16, 128, 2, 2, 31, 2, True, 3e-3, 1.2e-2)
    _depth_case("depth2_L14_T4", 224, 4, 14, 1024, 2, 16, 21, 1, True, 3e-3, 1.6e-2)


def test_errors_at_12_layers():
    """ViT-B/16, 12 layers (BASELINE configs[0] / [1] per-clip shape at 2 frames), forward and all 147 gradients."""
    _depth_case("depth12_B16_T2", 224, 2, 16, 768, 12, 12, 41, 1, True, 5e-3, 2.4e-2)


def test_errors_at_24_layers():
    """ViT-L/14, 24 layers: gradients of all 291 trainable tensors at 4 frames, and the full configs[3] per-clip
    shape (16 frames, 257 tokens per frame) forward."""
    _depth_case("depth24_L14_T4", 224, 4, 14, 1024, 24, 16, 51, 1, True, 8e-3, 3.2e-2)
    _depth_case("depth24_L14_T16_fwd", 224, 16, 14, 1024, 24, 16, 61, 1, False, 8e-3, 3.2e-2)


# ---- full-size properties --------------------------------------------------------------------------------------
def test_cfg3_shape_train_step_properties():
    """BASELINE configs[3] per-GPU shape: ViT-L/14 + AIM, 16 frames, 32 clips (the oracle cannot run this in seconds):
    two train steps through the recognizer; finite decreasing-or-equal loss is not asserted (random labels), but
    finiteness, the frozen set, non-zero gradients for all 291 + 2 trainable tensors and forward determinism are."""
    import aim_amd
    from aim_amd.dist import build_optimizer
    cfg = dict(type='Recognizer3D',
               backbone=dict(type='ViT_CLIP', input_resolution=224, patch_size=14, num_frames=16, width=1024, layers=24,
                             heads=16, drop_path_rate=0.2, adapter_scale=0.5, pretrained=None),
               cls_head=dict(type='I3DHead', in_channels=1024, num_classes=400, spatial_type='avg', dropout_ratio=0.5),
               test_cfg=dict(average_clips='prob'))
    torch.manual_seed(0)
    model = aim_amd.build_model(cfg)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "D_fc2" in n:
                p.normal_(0, 0.02)
    model = model.to(DEV).train()
    opt = build_optimizer(model, dict(type='AdamW', lr=3e-4, betas=(0.9, 0.999), weight_decay=0.05))
    gen = torch.Generator().manual_seed(5)
    imgs = torch.randn((32, 1, 3, 16, 224, 224), generator=gen).to(DEV)
    label = torch.randint(0, 400, (32, 1), generator=gen).to(DEV)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    assert len(trainable) == 24 * 12 + 3 + 2
    losses = []
    for _ in range(2):
        opt.zero_grad()
        loss = model(imgs, label, return_loss=True)["loss_cls"]
        loss.backward()
        if not losses:
            for n, p in model.named_parameters():
                if p.requires_grad:
                    assert torch.isfinite(p.grad).all() and p.grad.abs().max() > 0, n
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)), losses
    changed = sorted(n for n, p in model.named_parameters() if not torch.equal(p.detach(), before[n]))
    assert changed == trainable
    model.eval()
    with torch.no_grad():
        a = model.backbone(imgs[:4, 0])
        b = model.backbone(imgs[:4, 0])
    assert torch.equal(a, b) and torch.isfinite(a).all()
    _record("cfg3_shape_train_step", loss0=losses[0], loss1=losses[1], peak_gib=torch.cuda.max_memory_allocated() / 2 ** 30)


def test_eval_forward_keeps_no_block_context():
    """torch.no_grad() / eval forward must not keep the per-block saved tensors (needs_input_grad stays
    True under no_grad): peak memory of a no_grad forward stays far below the grad-enabled one."""
    m, _ = _model(224, 8, 16, 768, 12, 12, 7)
    imgs = torch.randn((8, 3, 8, 224, 224), generator=torch.Generator().manual_seed(3)).to(DEV)
    torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        y0 = m(imgs)
    torch.cuda.synchronize()
    peak_nograd = torch.cuda.max_memory_allocated() - base
    torch.cuda.reset_peak_memory_stats()
    y1 = m(imgs)
    torch.cuda.synchronize()
    peak_grad = torch.cuda.max_memory_allocated() - base
    assert torch.equal(y0, y1)
    assert y1.grad_fn is not None and y0.grad_fn is None
    assert peak_nograd < 0.4 * peak_grad, (peak_nograd, peak_grad)
    _record("eval_memory", peak_nograd_gib=peak_nograd / 2 ** 30, peak_grad_gib=peak_grad / 2 ** 30)


# ---- stream schedule as a race check ----------------------------------------------------------------------------
def _short_training(side: bool, B=8, steps=2):
    import bench
    from aim_amd import backbone as bb
    from aim_amd.dist import build_optimizer
    old = bb._USE_SIDE
    bb._USE_SIDE = side
    try:
        dev = torch.device("cuda", 0)
        torch.manual_seed(123)
        model = bench.build_model(8, dev)
        opt = build_optimizer(model, dict(type='AdamW', lr=3e-4, betas=(0.9, 0.999), weight_decay=0.05))
        g = torch.Generator(device="cpu").manual_seed(7)
        imgs = torch.randn((B, 1, 3, 8, 224, 224), generator=g).to(dev)
        label = torch.randint(0, 400, (B, 1), generator=g).to(dev)
        torch.manual_seed(99); torch.cuda.manual_seed(99)
        losses = []
        for _ in range(steps):
            opt.zero_grad()
            loss = model(imgs, label, return_loss=True)["loss_cls"]
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), opt.flat_p.detach().cpu().clone(), opt.flat_g.detach().cpu().clone()
    finally:
        bb._USE_SIDE = old


def test_side_streams_on_equals_off_bitwise():
    """Three overlapping HIP streams (AIM_SIDE_STREAM=1, the default) against everything inline on one stream: same
    kernels, same order per dependency chain, no atomics -> bit-identical losses, gradients and parameters.  A missing
    event / a buffer recycled across streams shows up as a difference."""
    l1, p1, g1 = _short_training(True)
    l0, p0, g0 = _short_training(False)
    assert torch.isfinite(l1).all()
    assert torch.equal(l1, l0), (l1, l0)
    assert torch.equal(g1, g0), (g1 - g0).abs().max()
    assert torch.equal(p1, p0), (p1 - p0).abs().max()


def test_training_overfits_a_fixed_batch():
    """End-to-end sanity of the whole train step (config -> Recognizer3D -> HIP backbone + head kernels -> FlatAdamW):
    on one fixed synthetic batch with fixed labels the loss must fall far below its initial ln(C) within 40 steps and the
    batch must end up classified correctly -- a sign or scaling error anywhere in the hand-written backward or the optimizer
    kernel shows up here even where a relative-error bound might be fooled."""
    import aim_amd
    from aim_amd.dist import build_optimizer
    cfg = dict(type='Recognizer3D',
               backbone=dict(type='ViT_CLIP', input_resolution=32, patch_size=16, num_frames=4, width=128, layers=3, heads=2,
                             drop_path_rate=0.0, adapter_scale=0.5, pretrained=None),
               cls_head=dict(type='I3DHead', in_channels=128, num_classes=7, spatial_type='avg', dropout_ratio=0.0),
               test_cfg=dict(average_clips='prob'))
    torch.manual_seed(0)
    model = aim_amd.build_model(cfg).to(DEV).train()
    opt = build_optimizer(model, dict(type='AdamW', lr=3e-3, betas=(0.9, 0.999), weight_decay=0.0))
    g = torch.Generator().manual_seed(1)
    imgs = torch.randn((8, 1, 3, 4, 32, 32), generator=g).to(DEV)
    label = torch.tensor([0, 1, 2, 3, 4, 5, 6, 0]).view(8, 1).to(DEV)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        out = model(imgs, label, return_loss=True)
        out["loss_cls"].backward()
        opt.all_reduce_grads()
        opt.step()
        losses.append(float(out["loss_cls"].detach()))
    assert abs(losses[0] - np.log(7)) < 0.3, losses[0]
    assert losses[-1] < 0.25 * losses[0], (losses[0], losses[-1])
    model.eval()
    with torch.no_grad():
        pred = np.concatenate([model(imgs[i:i + 1], return_loss=False) for i in range(8)]).argmax(1)
    assert (pred == label.view(-1).cpu().numpy()).all()
    _record("overfit_fixed_batch", loss0=losses[0], loss_last=losses[-1])
