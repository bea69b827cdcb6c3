"""End-to-end parity of the HIP backbone on a real MI355X.

Three comparators, in decreasing strictness of what they pin:
  * golden vectors produced by the REAL reference (fp32)             -> bf16-level tolerance
  * the CPU oracle with bf16 rounding at the product's rounding points -> 1e-3-class tolerance
    (BASELINE.json north_star: "within 1e-3 (bf16)"; see SURVEY section 7 hard part 1 for why the
    bf16 bar is taken against a same-rounding-points restatement)
  * size-independent properties at BASELINE config 2 size (determinism, frozen set, finiteness).
"""
import os

import numpy as np
import pytest
import torch

from oracle import vit_clip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _model(res, T, patch, D, L, H, seed, drop=0.0):
    import aim_amd
    m = aim_amd.ViT_CLIP(res, T, patch, D, L, H, drop)
    m.init_weights()
    st = O.synth_state_dict(O.backbone_param_shapes(res, T, patch, D, L), seed=seed)
    m.load_state_dict(st, strict=True)
    return m.to(DEV).eval(), st


def _relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _maxerr(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


@pytest.mark.parametrize("T", [2, 4])
def test_tiny_backbone_forward_backward(golden_dir, T):
    z = _load(golden_dir, f"backbone_tiny_T{T}.npz")
    D, H, L, B, T_, seed = [int(v) for v in z["meta"]]
    m, st = _model(32, T, 16, D, L, H, seed)
    y = m(z["imgs"].to(DEV))
    assert tuple(y.shape) == (B, D, T, 1, 1) and y.dtype == torch.float32
    # (1) same-rounding-point oracle: tight
    names = O.trainable_names(st)
    for n in names:
        st[n] = st[n].detach().requires_grad_(True)
    y_emu = O.emu_backbone(z["imgs"], st, H, rnd=O.BF16)
    # a value that lands on the other side of a bf16 rounding boundary moves by one bf16 ulp (2^-8
    # relative) and that propagates: the stable statistic is the relative L2 error (1e-3 class); the
    # max-abs bound is a few ulps of the O(1) outputs
    assert _relerr(y, y_emu) < 3e-3, _relerr(y, y_emu)
    assert _maxerr(y, y_emu) < 1.2e-2, _maxerr(y, y_emu)
    # (2) the real reference's fp32 output: bf16-level
    assert _maxerr(y, z["y"]) < 6e-2 and _relerr(y, z["y"]) < 1.5e-2, (_maxerr(y, z["y"]), _relerr(y, z["y"]))
    # gradients of the full trainable set against the reference's autograd
    y.backward(z["g"].to(DEV))
    got = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    assert sorted(got) == sorted(names) and all(g is not None for g in got.values())
    worst = 0.0
    for n in names:
        ref = z["grad." + n]
        assert got[n].shape == ref.shape
        e = _relerr(got[n], ref)
        worst = max(worst, e)
        assert e < 2.5e-2, (n, e)      # measured <= 1.4e-2 (gpurun_out/parity_r02.jsonl)
    # frozen tensors received no gradient
    assert all(p.grad is None for n, p in m.named_parameters() if not p.requires_grad)


@pytest.mark.parametrize("T", [2, 4])
def test_block_against_reference_fixture(golden_dir, T):
    """One ResidualAttentionBlock on the reference's own input (vit_clip.py:199-288)."""
    from aim_amd import backbone as bb
    z = _load(golden_dir, f"block_tiny_T{T}.npz")
    D, H, N, B, T_, seed = [int(v) for v in z["meta"]]
    m, st = _model(32, T, 16, D, 1, H, seed)
    fz = m._frozen_operands()["blocks"][0]
    blk = m.transformer.resblocks[0]
    adp = {a: bb._AdapterW(getattr(blk, a).D_fc1.weight, getattr(blk, a).D_fc1.bias, getattr(blk, a).D_fc2.weight,
                           getattr(blk, a).D_fc2.bias) for a in bb._ADAPTERS if a != "MLP_Adapter"}   # stand-alone casts
    ma = blk.MLP_Adapter
    fz.stage_mlp_adapter(ma.D_fc1.weight, ma.D_fc1.bias, ma.D_fc2.weight, ma.D_fc2.bias)
    x = z["x"].permute(1, 0, 2).contiguous().reshape(B * T * N, D).to(DEV)
    dms = torch.full((N,), 0.5, device=DEV)
    y, c = bb._block_forward(x, fz, adp, B, T, N, H, dms, dms, True)
    y_ref = z["y"].permute(1, 0, 2).reshape(B * T * N, D)
    assert _relerr(y, y_ref) < 8e-3, _relerr(y, y_ref)
    assert _relerr(c["lam"], z["lamda"]) < 1e-2, _relerr(c["lam"], z["lamda"])   # exp() of bf16-rounded q.k
    x_e = z["x"].permute(1, 0, 2).contiguous()
    y_emu, aux = O.emu_block(x_e, st, 0, H, T, 0.5, O.BF16, return_aux=True)
    assert _relerr(y, y_emu.reshape(-1, D)) < 1e-3, _relerr(y, y_emu.reshape(-1, D))          # 1e-3 (bf16) bar
    assert _maxerr(y, y_emu.reshape(-1, D)) < 8e-3, _maxerr(y, y_emu.reshape(-1, D))          # <= 2 bf16 ulps at |x|~3
    assert _relerr(c["lam"], aux["lamda"]) < 1e-3
    # backward: dX and the 12 adapter gradients
    def _param(a, leaf):
        mod, attr = leaf.split(".")
        return getattr(getattr(getattr(blk, a), mod), attr)
    grads = {a: {leaf: torch.zeros_like(_param(a, leaf), dtype=torch.float32) for leaf in bb._ADAPTER_LEAVES}
             for a in bb._ADAPTERS}
    g = z["g"].permute(1, 0, 2).contiguous().reshape(B * T * N, D).to(DEV)
    dx = bb._block_backward(g.to(torch.bfloat16), c, fz, adp, grads, B, T, N, H)
    dx_ref = z["dx"].permute(1, 0, 2).reshape(B * T * N, D)
    assert _relerr(dx, dx_ref) < 1.5e-2, _relerr(dx, dx_ref)       # measured 5e-3
    for a in bb._ADAPTERS:
        for leaf in bb._ADAPTER_LEAVES:
            e = _relerr(grads[a][leaf], z[f"grad.{a}.{leaf}"])
            assert e < 2.5e-2, (a, leaf, e)                              # measured <= 1.1e-2


def test_cfg1_shape_forward(golden_dir):
    """BASELINE.json configs[0] (ViT-B/16 + AIM, 2 frames 224^2, batch 1) on the HIP path."""
    z = _load(golden_dir, "backbone_cfg1.npz")
    D, H, L, B, T, seed = [int(v) for v in z["meta"]]
    m, st = _model(224, T, 16, D, L, H, seed)
    imgs = torch.randn((1, 3, T, 224, 224), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        y = m(imgs.to(DEV))
        y_emu = O.emu_backbone(imgs, st, H, rnd=O.BF16)
    ref_drift = _maxerr(z["y_autocast_bf16"], z["y"])
    # vs the real reference (fp32): no worse than the reference's own bf16-autocast drift
    assert _maxerr(y, z["y"]) < max(5e-2, 2 * ref_drift), (_maxerr(y, z["y"]), ref_drift)
    # vs same-rounding-point oracle
    assert _relerr(y, y_emu) < 5e-3, _relerr(y, y_emu)


def test_recognizer_from_config_class_indices(golden_dir):
    """Registry / config surface: build Recognizer3D(ViT_CLIP + I3DHead) from a config dict shaped like
    configs/_base_/models/vitclip_base.py; predicted class indices match the oracle bit-exactly
    wherever the oracle's top-1 margin exceeds the bf16 tolerance."""
    import aim_amd
    z = _load(golden_dir, "backbone_tiny_T4.npz")
    D, H, L, B, T, seed = [int(v) for v in z["meta"]]
    cfg = aim_amd.Config.fromfile(os.path.join(os.path.dirname(__file__), "data", "vitclip_tiny_cfg.py"))
    # head dropout off (config override, as --cfg-options would): the train-mode loss is then deterministic and is
    # pinned against the golden loss_cls at bf16 tolerance
    cfg.merge_from_dict({"model.backbone.num_frames": T, "model.cls_head.dropout_ratio": 0.0})
    model = aim_amd.build_model(cfg.model).to(DEV).eval()
    st = O.synth_state_dict(O.backbone_param_shapes(32, T, 16, D, L), seed=seed)
    model.backbone.load_state_dict(st, strict=True)
    C = z["fc_w"].shape[0]
    with torch.no_grad():
        model.cls_head.fc_cls.weight.copy_(z["fc_w"]); model.cls_head.fc_cls.bias.copy_(z["fc_b"])
    imgs = z["imgs"].unsqueeze(1).to(DEV)            # [B, 1, 3, T, H, W]
    # max_testing_views (test_cfg) requires batch 1 per call, as in the reference (recognizer3d.py:40-42)
    with torch.no_grad():
        out = np.concatenate([model(imgs[i:i + 1], return_loss=False) for i in range(B)])   # [B, C] probabilities
        with pytest.raises(AssertionError, match="batch_size == 1"):
            model(imgs, return_loss=False)
    ref_score = z["cls_score"]
    ref_prob = torch.softmax(ref_score, 1)
    assert out.shape == (B, C)
    assert np.abs(out - ref_prob.numpy()).max() < 2e-2
    top2 = ref_score.topk(2, dim=1).values
    margin_ok = (top2[:, 0] - top2[:, 1]) > 5e-2
    assert margin_ok.any()
    pred = torch.from_numpy(out).argmax(1)
    assert torch.equal(pred[margin_ok], z["pred"].long()[margin_ok])
    # training interface: losses dict + one backward
    model.train()
    losses = model(imgs, z["label"].long().view(B, 1).to(DEV), return_loss=True)
    assert set(losses) == {"top1_acc", "top5_acc", "loss_cls"}
    loss, log_vars = model._parse_losses(losses)
    loss.backward()
    assert model.cls_head.dropout is None
    assert abs(log_vars["loss_cls"] - float(z["loss_cls"])) < 2e-2, (log_vars["loss_cls"], float(z["loss_cls"]))
    assert all(p.grad is not None for p in model.parameters() if p.requires_grad)


def test_droppath_mask_semantics():
    """DropPath drops TOKEN positions across the whole batch (timm mask shape (x.shape[0],1,1) on the
    reference's [N,BT,D] tensor), scaled by 1/keep, times adapter_scale; identity in eval."""
    import aim_amd
    torch.manual_seed(0)
    m = aim_amd.ViT_CLIP._drop_mask(1000, 0.25, 0.5, True, DEV)
    vals = sorted(m.unique().cpu().tolist())
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 0.5 / 0.75) < 1e-6
    assert abs((m > 0).float().mean().item() - 0.75) < 0.05
    e = aim_amd.ViT_CLIP._drop_mask(7, 0.25, 0.5, False, DEV)
    assert torch.equal(e, torch.full((7,), 0.5, device=DEV))
    model = aim_amd.ViT_CLIP(32, 2, 16, 128, 4, 2, 0.3)
    dpr = [b.drop_prob for b in model.transformer.resblocks]
    assert np.allclose(dpr, np.linspace(0, 0.3, 4))
    # the whole model's factors in one draw: [L, 2, N], per layer {0, scale / keep} at its own rate, two independent rows
    mk = model._drop_masks(4000, True, torch.device(DEV))
    assert tuple(mk.shape) == (4, 2, 4000) and torch.equal(mk[0], torch.full((2, 4000), 0.5, device=DEV))   # adapter_scale 0.5, rate 0
    for l in (1, 2, 3):
        keep = 1 - dpr[l]
        vals = sorted(mk[l].unique().cpu().tolist())
        assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 0.5 / keep) < 1e-6
        assert abs((mk[l] > 0).float().mean().item() - keep) < 0.03
        assert not torch.equal(mk[l, 0], mk[l, 1])
    assert torch.equal(model._drop_masks(7, False, torch.device(DEV)), torch.full((4, 2, 7), 0.5, device=DEV))
    # the per-layer constants are cached on the device (no host-to-device copy per step) and follow the blocks' attributes
    consts = model._drop_consts
    mk2 = model._drop_masks(4000, True, torch.device(DEV))
    assert model._drop_consts is consts and not torch.equal(mk, mk2)          # same constants, a new draw
    model.transformer.resblocks[3].drop_prob = 0.5
    mk3 = model._drop_masks(4000, True, torch.device(DEV))
    assert model._drop_consts is not consts
    vals = sorted(mk3[3].unique().cpu().tolist())
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 0.5 / 0.5) < 1e-6


def test_full_size_properties():
    """BASELINE configs[1] exactly (ViT-B/16, 8 frames, 224^2, 64 clips on one GPU), which the CPU oracle
    cannot check in seconds: bitwise determinism of the forward, finite outputs, every trainable
    tensor gets a finite non-zero gradient, and one AdamW step changes only the trainable set."""
    m, st = _model(224, 8, 16, 768, 12, 12, 7)
    m.train()   # drop_path_rate 0 here, so train == eval numerically
    B = 64
    imgs = torch.randn((B, 3, 8, 224, 224), generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        y1 = m(imgs)
        y2 = m(imgs)
    assert torch.equal(y1, y2) and torch.isfinite(y1).all()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    y = m(imgs)
    (y * torch.randn_like(y)).sum().backward()
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0, n
    opt.step()
    changed = sorted(n for n, p in m.named_parameters() if not torch.equal(p.detach(), before[n]))
    assert changed == sorted(O.trainable_names(st))
    # linearity of the backward in the upstream gradient (size-independent property)
    g1 = torch.randn_like(y)
    opt.zero_grad(set_to_none=True)
    ga = torch.autograd.grad(m(imgs), m.temporal_embedding, g1)[0]
    gb = torch.autograd.grad(m(imgs), m.temporal_embedding, 2 * g1)[0]
    assert ((gb - 2 * ga).norm() / gb.norm()).item() < 2e-2


def test_vit_l14_shape_matches_oracle():
    """ViT-L/14 geometry (BASELINE configs[3]/[4]: patch 14, width 1024, 16 heads, N = 257 tokens,
    3*14*14 = 588 patch columns padded to 640) at 2 layers / 4 frames: forward + adapter gradients."""
    T, D, L, H = 4, 1024, 2, 16
    m, st = _model(224, T, 14, D, L, H, 21)
    imgs = torch.randn((1, 3, T, 224, 224), generator=torch.Generator().manual_seed(8))
    g = torch.randn((1, D, T, 1, 1), generator=torch.Generator().manual_seed(9))
    y = m(imgs.to(DEV))
    names = O.trainable_names(st)
    for n in names:
        st[n] = st[n].detach().requires_grad_(True)
    y_emu = O.emu_backbone(imgs, st, H, rnd=O.BF16)
    assert tuple(y.shape) == (1, D, T, 1, 1)
    assert _relerr(y, y_emu) < 3e-3, _relerr(y, y_emu)
    y.backward(g.to(DEV))
    grads = torch.autograd.grad(y_emu, [st[n] for n in names], g)
    got = dict(m.named_parameters())
    for n, gr in zip(names, grads):
        e = _relerr(got[n].grad, gr)
        assert e < 2.5e-2, (n, e)


def test_uint8_input_with_fused_gpu_normalize(golden_dir):
    """GPUNormalize pre-hook (module_hooks.py:35-87) fused into the patch gather: uint8 clip in, same
    features as normalising in PyTorch first."""
    import aim_amd
    z = _load(golden_dir, "backbone_tiny_T2.npz")
    D, H, L, B, T, seed = [int(v) for v in z["meta"]]
    cfg = dict(type='Recognizer3D',
               backbone=dict(type='ViT_CLIP', input_resolution=32, patch_size=16, num_frames=T, width=D, layers=L,
                             heads=H, drop_path_rate=0.0),
               cls_head=dict(type='I3DHead', in_channels=D, num_classes=5, dropout_ratio=0.0),
               test_cfg=dict(average_clips='prob'))
    model = aim_amd.build_model(cfg).to(DEV).eval()
    st = O.synth_state_dict(O.backbone_param_shapes(32, T, 16, D, L), seed=seed)
    model.backbone.load_state_dict(st, strict=True)
    mean, std = [122.769, 116.74, 104.04], [68.493, 66.63, 70.321]
    u8 = (torch.rand((B, 3, T, 32, 32), generator=torch.Generator().manual_seed(4)) * 255).to(torch.uint8)
    with torch.no_grad():
        ref = model.backbone(O.ref_gpu_normalize(u8, mean, std).to(DEV))
        handles = aim_amd.register_module_hooks(model, [dict(type='GPUNormalize', hooked_module='backbone',
                                                             hook_pos='forward_pre', input_format='NCTHW',
                                                             mean=mean, std=std)])
        out = model.backbone(u8.to(DEV))
    assert len(handles) == 1
    assert _relerr(out, ref) < 2e-3, _relerr(out, ref)
    with pytest.raises(AssertionError, match="uint8"):
        model.backbone(u8.float().to(DEV))
    # the fused normalisation is armed per call by the hook: once the hook is gone, an already-normalised float clip is
    # NOT normalised again, and a uint8 clip without a hook is refused
    for h in handles:
        h.remove()
    with torch.no_grad():
        again = model.backbone(O.ref_gpu_normalize(u8, mean, std).to(DEV))
        assert torch.equal(again, ref)
        with pytest.raises(TypeError, match="GPUNormalize"):
            model.backbone(u8.to(DEV))
