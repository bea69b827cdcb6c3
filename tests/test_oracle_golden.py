"""Pin the CPU oracle against the golden vectors produced by the real reference
(tests/golden/make_golden.py).  fp32 bar: 1e-5 (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from oracle import vit_clip_oracle as O

TOL = 1e-5


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k] for k in z.files}


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float32)


def _close(a, b, tol=TOL, rel=0.0):
    a, b = a.double(), b.double()
    err = (a - b).abs().max().item()
    bound = tol + rel * b.abs().max().item()
    assert err <= bound, f"max abs err {err:.3e} > {bound:.3e}"


@pytest.mark.parametrize("T", [2, 4])
def test_ref_block_forward_backward(golden_dir, T):
    z = _load(golden_dir, f"block_tiny_T{T}.npz")
    D, H, N, B, T_, seed = [int(v) for v in z["meta"]]
    st = O.synth_state_dict(O.backbone_param_shapes(32, T, 16, D, 1), seed=seed)
    for n in O.trainable_names(st):
        st[n].requires_grad_(True)
    x = z["x"].clone().requires_grad_(True)
    y, aux = O.ref_block(x, st, 0, H, T, 0.5, return_aux=True)
    _close(y, z["y"])
    _close(aux["ow"], z["ow"], rel=1e-6)
    _close(aux["cw"], z["cw"], rel=1e-6)
    _close(aux["lamda"], z["lamda"], 1e-6)
    _close(aux["xt"], z["xt"])
    names = [k[5:] for k in z if k.startswith("grad.")]
    assert len(names) == 12
    full = ["transformer.resblocks.0." + n for n in names]
    grads = torch.autograd.grad(y, [x] + [st[n] for n in full], z["g"])
    _close(grads[0], z["dx"], 2e-5)
    for n, g in zip(names, grads[1:]):
        _close(g, z["grad." + n], 1e-5, rel=1e-5)


@pytest.mark.parametrize("T", [2, 4])
def test_emu_block_matches_reference(golden_dir, T):
    """Product dataflow (frame-major, de-duplicated algebra) == reference in fp32."""
    z = _load(golden_dir, f"block_tiny_T{T}.npz")
    D, H, N, B, T_, seed = [int(v) for v in z["meta"]]
    st = O.synth_state_dict(O.backbone_param_shapes(32, T, 16, D, 1), seed=seed)
    for n in O.trainable_names(st):
        st[n].requires_grad_(True)
    x = z["x"].permute(1, 0, 2).contiguous().requires_grad_(True)       # [BT,N,D]
    y, aux = O.emu_block(x, st, 0, H, T, 0.5, O.FP32, return_aux=True)
    _close(y.permute(1, 0, 2), z["y"], 2e-5)
    _close(aux["lamda"], z["lamda"], 1e-6)
    _close(aux["xt"], z["xt"], 2e-5)
    names = [k[5:] for k in z if k.startswith("grad.")]
    full = ["transformer.resblocks.0." + n for n in names]
    grads = torch.autograd.grad(y, [x] + [st[n] for n in full], z["g"].permute(1, 0, 2))
    _close(grads[0].permute(1, 0, 2), z["dx"], 5e-5)
    for n, g in zip(names, grads[1:]):
        _close(g, z["grad." + n], 2e-5, rel=2e-5)


def test_ref_block_real_shape(golden_dir):
    z = _load(golden_dir, "block_real_T2.npz")
    D, H, N, B, T, seed = [int(v) for v in z["meta"]]
    st = O.synth_state_dict(O.backbone_param_shapes(224, T, 16, D, 1), seed=seed)
    x = _randn((N, B * T, D), seed + 1)
    with torch.no_grad():
        y, aux = O.ref_block(x, st, 0, H, T, 0.5, return_aux=True)
        y2, aux2 = O.emu_block(x.permute(1, 0, 2).contiguous(), st, 0, H, T, 0.5, O.FP32, return_aux=True)
    _close(y.reshape(-1)[z["idx"].long()], z["y_sample"], 2e-5)
    assert abs(y.double().sum().item() - float(z["y_sum"])) < 1e-2
    assert abs((y.double() ** 2).sum().item() / float(z["y_sq"]) - 1) < 1e-6
    _close(aux["lamda"], z["lamda"], 1e-6)
    _close(aux["ow"], z["ow"], rel=1e-5)
    _close(aux["cw"], z["cw"], rel=1e-5)
    _close(y2.permute(1, 0, 2).reshape(-1)[z["idx"].long()], z["y_sample"], 5e-5)
    _close(aux2["lamda"], z["lamda"], 1e-6)


@pytest.mark.parametrize("T", [2, 4])
def test_backbone_tiny(golden_dir, T):
    z = _load(golden_dir, f"backbone_tiny_T{T}.npz")
    D, H, L, B, T_, seed = [int(v) for v in z["meta"]]
    st = O.synth_state_dict(O.backbone_param_shapes(32, T, 16, D, L), seed=seed)
    names = O.trainable_names(st)
    assert len(names) == 12 * L + 3
    for fn in (lambda: O.ref_backbone(z["imgs"], st, H, T), lambda: O.emu_backbone(z["imgs"], st, H)):
        for n in names:
            st[n] = st[n].detach().requires_grad_(True)
        y = fn()
        assert tuple(y.shape) == (B, D, T, 1, 1)
        _close(y, z["y"], 2e-5)
        grads = torch.autograd.grad(y, [st[n] for n in names], z["g"])
        for n, g in zip(names, grads):
            _close(g, z["grad." + n], 2e-5, rel=5e-5)
    # recognizer-level pins (head, CE loss, class indices bit-exact)
    score = O.ref_i3d_head(z["y"], z["fc_w"], z["fc_b"])
    _close(score, z["cls_score"], 1e-6)
    assert torch.equal(score.argmax(1), z["pred"].long())
    _close(O.ref_cross_entropy(score, z["label"].long()), z["loss_cls"], 1e-6)


def test_backbone_cfg1(golden_dir):
    """BASELINE.json configs[0]: ViT-B/16 + AIM, 2 frames 224^2, batch 1, fp32 CPU forward."""
    z = _load(golden_dir, "backbone_cfg1.npz")
    D, H, L, B, T, seed = [int(v) for v in z["meta"]]
    st = O.synth_state_dict(O.backbone_param_shapes(224, T, 16, D, L), seed=seed)
    imgs = _randn((1, 3, T, 224, 224), 2)
    with torch.no_grad():
        y = O.ref_backbone(imgs, st, H, T)
        y2 = O.emu_backbone(imgs, st, H)
        y3 = O.emu_backbone(imgs, st, H, rnd=O.BF16)
    _close(y, z["y"], 2e-5)
    _close(y2, z["y"], 1e-4)
    # bf16 rounding-point emulation stays within bf16-level distance of the fp32 reference,
    # comparable to the reference's own autocast drift (hard part 1)
    ref_drift = (z["y_autocast_bf16"] - z["y"]).abs().max().item()
    emu_drift = (y3 - z["y"]).abs().max().item()
    assert emu_drift < max(5e-2, 3 * ref_drift), (emu_drift, ref_drift)


def test_top_k_and_average_clip():
    s = np.array([[0.1, 0.5, 0.4], [0.3, 0.3, 0.4], [0.9, 0.05, 0.05]])
    assert O.ref_top_k_accuracy(s, [1, 0, 0], (1, 2)) == [2 / 3, 2 / 3]
    cs = torch.tensor(s, dtype=torch.float32)
    out = O.ref_average_clip(cs, 3, "prob")
    assert out.shape == (1, 3)
    _close(out, torch.softmax(cs, 1).mean(0, keepdim=True), 1e-7)


def test_synth_weights_are_deterministic():
    sh = O.backbone_param_shapes(32, 2, 16, 128, 1)
    a, b = O.synth_state_dict(sh, 7), O.synth_state_dict(sh, 7)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["conv1.weight"], O.synth_state_dict(sh, 8)["conv1.weight"])


# ---- DropPath active (train mode): fixtures hold the masks the REAL reference drew (vit_clip.py:112,275,286) ----
def _sample_close(t, z, key, tol, rel=0.0):
    flat = t.detach().reshape(-1)
    _close(flat[z[key + ".idx"].long()], z[key + ".val"], tol, rel)
    assert abs((flat.double() ** 2).sum().item() / max(float(z[key + ".sq"]), 1e-30) - 1) < 1e-4, key


@pytest.mark.parametrize("T", [2, 4])
def test_block_droppath_tiny(golden_dir, T):
    """ref_block / emu_block with the reference's two masks (zeros included, m1 != m2): y, dX, 12 adapter grads."""
    z = _load(golden_dir, f"block_tiny_T{T}_droppath.npz")
    D, H, N, B, T_, seed = [int(v) for v in z["meta"]]
    m1, m2 = z["m1"], z["m2"]
    assert (m1 == 0).any() and (m2 == 0).any() and not torch.equal(m1, m2)
    assert set(np.unique(m1.numpy())) <= {0.0, np.float32(1 / (1 - float(z["rate"])))}
    st = O.synth_state_dict(O.backbone_param_shapes(32, T, 16, D, 2), seed=seed)
    names = [k[5:] for k in z if k.startswith("grad.")]
    assert len(names) == 12
    full = ["transformer.resblocks.1." + n for n in names]
    for n in full:
        st[n].requires_grad_(True)
    x = z["x"].clone().requires_grad_(True)
    y = O.ref_block(x, st, 1, H, T, 0.5, drop_mask=(m1, m2))
    _close(y, z["y"])
    grads = torch.autograd.grad(y, [x] + [st[n] for n in full], z["g"])
    _close(grads[0], z["dx"], 2e-5)
    for n, g in zip(names, grads[1:]):
        _close(g, z["grad." + n], 1e-5, rel=1e-5)
    xe = z["x"].permute(1, 0, 2).contiguous().requires_grad_(True)
    ye = O.emu_block(xe, st, 1, H, T, 0.5, O.FP32, drop_mask=(m1, m2))
    _close(ye.permute(1, 0, 2), z["y"], 2e-5)
    ge = torch.autograd.grad(ye, [xe] + [st[n] for n in full], z["g"].permute(1, 0, 2))
    _close(ge[0].permute(1, 0, 2), z["dx"], 5e-5)
    for n, g in zip(names, ge[1:]):
        _close(g, z["grad." + n], 2e-5, rel=2e-5)
    # the masks matter: a single shared mask gives a different block output
    assert (O.ref_block(z["x"], st, 1, H, T, 0.5, drop_mask=m1).detach() - z["y"]).abs().max() > 1e-3


def test_block_droppath_real_shape(golden_dir):
    z = _load(golden_dir, "block_real_T2_droppath.npz")
    D, H, N, B, T, seed = [int(v) for v in z["meta"]]
    st = O.synth_state_dict(O.backbone_param_shapes(224, T, 16, D, 2), seed=seed)
    names = sorted(k[5:-4] for k in z if k.startswith("grad.") and k.endswith(".idx"))
    assert len(names) == 12
    full = ["transformer.resblocks.1." + n for n in names]
    for n in full:
        st[n].requires_grad_(True)
    x = _randn((N, B * T, D), seed + 1).requires_grad_(True)
    g = _randn((N, B * T, D), seed + 2)
    y = O.ref_block(x, st, 1, H, T, 0.5, drop_mask=(z["m1"], z["m2"]))
    _sample_close(y, z, "y", 2e-5)
    grads = torch.autograd.grad(y, [x] + [st[n] for n in full], g)
    _sample_close(grads[0], z, "dx", 2e-5)
    for n, gr in zip(names, grads[1:]):
        _sample_close(gr, z, "grad." + n, 2e-5, rel=1e-4)


def test_backbone_droppath_tiny(golden_dir):
    z = _load(golden_dir, "backbone_tiny_T2_droppath.npz")
    D, H, L, B, T, seed = [int(v) for v in z["meta"]]
    st = O.synth_state_dict(O.backbone_param_shapes(32, T, 16, D, L), seed=seed)
    names = O.trainable_names(st)
    mk = z["masks"]
    masks = [None, (mk[0], mk[1]), (mk[2], mk[3])]      # layer 0 has rate 0 (vit_clip.py:297)
    for fn in (lambda: O.ref_backbone(z["imgs"], st, H, T, drop_masks=masks),
               lambda: O.emu_backbone(z["imgs"], st, H, drop_masks=masks)):
        for n in names:
            st[n] = st[n].detach().requires_grad_(True)
        y = fn()
        _close(y, z["y"], 2e-5)
        grads = torch.autograd.grad(y, [st[n] for n in names], z["g"])
        for n, g in zip(names, grads):
            _close(g, z["grad." + n], 2e-5, rel=5e-5)


# ---- stock-AIM variant (mmaction/models/backbones/vitclip_aim.py, class AIM, wind_attn=False): SURVEY 8f-4 -----------
@pytest.mark.parametrize("name,train", [("aim_backbone_tiny_T2.npz", False), ("aim_backbone_tiny_T4_droppath.npz", True)])
def test_aim_backbone_tiny(golden_dir, name, train):
    z = _load(golden_dir, name)
    D, H, L, B, T, seed = [int(v) for v in z["meta"]]
    st = O.synth_state_dict(O.backbone_param_shapes(32, T, 16, D, L), seed=seed)
    names = O.trainable_names(st)
    masks = None
    if train:
        mk = z["masks"]
        assert any((k == 0).any() for k in mk)
        masks = [None, (mk[0], mk[1]), (mk[2], mk[3])]
    for fn in (lambda: O.ref_aim_backbone(z["imgs"], st, H, T, drop_masks=masks),
               lambda: O.emu_aim_backbone(z["imgs"], st, H, drop_masks=masks)):
        for n in names:
            st[n] = st[n].detach().requires_grad_(True)
        y = fn()
        assert tuple(y.shape) == (B, D, T, 1, 1)
        _close(y, z["y"], 2e-5)
        grads = torch.autograd.grad(y, [st[n] for n in names], z["g"])
        for n, g in zip(names, grads):
            _close(g, z["grad." + n], 2e-5, rel=5e-5)
    # it is a different network from vit_clip.py's block on the same weights
    assert (O.ref_backbone(z["imgs"], st, H, T).detach() - z["y"]).abs().max() > 1e-2
