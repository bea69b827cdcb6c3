#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference).  It loads
``mmaction/models/backbones/vit_clip.py`` by path after pre-seeding
``sys.modules`` with four tiny stand-ins for packages that are absent here
(timm.models.layers, clip, mmaction.utils, mmaction.models.builder) --
SURVEY.md section 8(c).  No reference source is copied: only inputs (or their
seeds) and the reference's numeric outputs are stored.

    python tests/golden/make_golden.py

Weights come from ``oracle.vit_clip_oracle.synth_state_dict`` (name-seeded, so
the same tensors can be rebuilt on the GPU box without the reference).
"""
import importlib.util
import logging
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import vit_clip_oracle as O  # noqa: E402

REF = "/root/reference/mmaction/models/backbones/vit_clip.py"
REF_AIM = "/root/reference/mmaction/models/backbones/vitclip_aim.py"


def load_reference():
    def stub(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    stub("timm"); stub("timm.models")
    tl = stub("timm.models.layers")

    class DropPath(nn.Module):  # timm 0.5.4 semantics; only exercised in train mode
        drawn = []              # every mask drawn (factor bernoulli/keep, shape [x.shape[0]]), in call order

        def __init__(self, drop_prob=0.):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0. or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            mask = mask.div_(keep)
            DropPath.drawn.append(mask.reshape(-1).clone())
            return x * mask

    tl.DropPath = DropPath
    tl.to_2tuple = lambda x: (x, x)
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    stub("clip")
    stub("mmaction")
    stub("mmaction.utils").get_root_logger = lambda *a, **k: logging.getLogger("ref")
    stub("mmaction.models"); stub("mmaction.models.backbones")

    class _Reg:
        def register_module(self, *a, **k):
            return lambda c: c

    stub("mmaction.models.builder").BACKBONES = _Reg()
    spec = importlib.util.spec_from_file_location("mmaction.models.backbones.vit_clip", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_aim():
    """The stock-AIM file (class ``AIM``); same stand-ins as ``load_reference`` (call that first)."""
    spec = importlib.util.spec_from_file_location("mmaction.models.backbones.vitclip_aim", REF_AIM)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def gen_aim(mod_aim, T, seed, train):
    """Whole tiny stock-AIM backbone (``AIM(..., wind_attn=False)``, 3 layers): output + all trainable gradients; in
    train mode with the DropPath masks the reference drew (two per block with rate > 0)."""
    D, H, L, B = 128, 2, 3, 2
    logging.getLogger("ref").setLevel(logging.ERROR)
    m = mod_aim.AIM(32, T, 16, D, L, H, drop_path_rate=0.5 if train else 0.0, adapter_scale=0.5, wind_attn=False)
    m.init_weights()
    st = O.synth_state_dict(O.backbone_param_shapes(32, T, 16, D, L), seed=seed)
    msg = m.load_state_dict(st, strict=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    m.train() if train else m.eval()
    imgs = randn((B, 3, T, 32, 32), seed + 1)
    g = randn((B, D, T, 1, 1), seed + 2)
    torch.manual_seed(seed + 9)
    del _drawn()[:]
    y = m(imgs)
    masks = list(_drawn())
    params = {n: p for n, p in m.named_parameters() if p.requires_grad}
    assert sorted(params) == sorted(O.trainable_names(st))
    grads = torch.autograd.grad(y, list(params.values()), g)
    out = dict(imgs=imgs, g=g, y=y, meta=np.array([D, H, L, B, T, seed]))
    if train:
        assert len(masks) == 4 and any((k == 0).any() for k in masks)
        out["masks"] = torch.stack(masks)
    for (n, _), gr in zip(params.items(), grads):
        out["grad." + n] = gr
    np.savez_compressed(os.path.join(HERE, f"aim_backbone_tiny_T{T}{'_droppath' if train else ''}.npz"), **npify(out))


def randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float32)


def build_ref(mod, res, T, patch, width, layers, heads, seed, drop_path_rate=0.0):
    logging.getLogger("ref").setLevel(logging.ERROR)
    m = mod.ViT_CLIP(res, T, patch, width, layers, heads, drop_path_rate=drop_path_rate, adapter_scale=0.5)
    m.init_weights()  # applies the freeze policy
    st = O.synth_state_dict(O.backbone_param_shapes(res, T, patch, width, layers), seed=seed)
    missing = m.load_state_dict(st, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    m.eval()
    return m, st


def npify(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def gen_block(mod, T, seed):
    D, H, N, B = 128, 2, 5, 2
    m, st = build_ref(mod, 32, T, 16, D, 1, H, seed)
    blk = m.transformer.resblocks[0]
    x = randn((N, B * T, D), seed + 1).requires_grad_(True)
    g = randn((N, B * T, D), seed + 2)
    y = blk(x)
    params = {n: p for n, p in blk.named_parameters() if p.requires_grad}
    grads = torch.autograd.grad(y, [x] + list(params.values()), g)
    with torch.no_grad():
        xl = blk.ln_1(x)
        ct = x[:1].reshape(1, B, T, D).permute(2, 1, 0, 3).reshape(T, B, D)
        xt = blk.T_Adapter(blk.attention(blk.ln_1(ct)))
        xt = xt.reshape(T, B, 1, D).permute(2, 1, 0, 3).reshape(1, B * T, D)
        _, ow = blk.attention(xl, need_weights=True)
        _, cw = blk.cross_attention(xl, xt, need_weights=True)
    out = dict(x=x, g=g, y=y, ow=ow, cw=cw, lamda=cw / (cw + ow), xt=xt[0], dx=grads[0],
               meta=np.array([D, H, N, B, T, seed]))
    for (n, _), gr in zip(params.items(), grads[1:]):
        out["grad." + n] = gr
    np.savez_compressed(os.path.join(HERE, f"block_tiny_T{T}.npz"), **npify(out))


def gen_block_real(mod, seed):
    D, H, N, B, T = 768, 12, 197, 1, 2
    m, st = build_ref(mod, 224, T, 16, D, 1, H, seed)
    blk = m.transformer.resblocks[0]
    x = randn((N, B * T, D), seed + 1)
    with torch.no_grad():
        y = blk(x)
        xl = blk.ln_1(x)
        ct = x[:1].reshape(1, B, T, D).permute(2, 1, 0, 3).reshape(T, B, D)
        xt = blk.T_Adapter(blk.attention(blk.ln_1(ct)))
        xt = xt.reshape(T, B, 1, D).permute(2, 1, 0, 3).reshape(1, B * T, D)
        _, ow = blk.attention(xl, need_weights=True)
        _, cw = blk.cross_attention(xl, xt, need_weights=True)
    idx = torch.randperm(y.numel(), generator=torch.Generator().manual_seed(seed + 3))[:8192]
    out = dict(idx=idx, y_sample=y.reshape(-1)[idx], y_sum=y.double().sum(), y_abs=y.double().abs().sum(),
               y_sq=(y.double() ** 2).sum(), ow=ow, cw=cw, lamda=cw / (cw + ow), xt=xt[0],
               meta=np.array([D, H, N, B, T, seed]))
    np.savez_compressed(os.path.join(HERE, "block_real_T2.npz"), **npify(out))


def _drawn():
    import sys as _s
    return _s.modules["timm.models.layers"].DropPath.drawn


def _sample(t, k, seed):
    """Fixture-sized view of a big tensor: k sampled elements + sum / sum of squares (fp64)."""
    flat = t.detach().reshape(-1)
    idx = torch.randperm(flat.numel(), generator=torch.Generator().manual_seed(seed))[:k]
    return dict(idx=idx, val=flat[idx], sum=flat.double().sum(), sq=(flat.double() ** 2).sum())


def gen_block_droppath(mod, T, seed, real=False):
    """One block in TRAIN mode with DropPath active (vit_clip.py:112,275,286): the second block of a 2-layer
    model (layer 0 has rate 0 = Identity, :297).  The masks the reference drew are stored with its outputs, so the
    product and the oracle can be fed the same masks.  rate 0.5 at N = 5 / 0.3 at N = 197: zeros occur."""
    if real:
        D, H, N, B, res, rate = 768, 12, 197, 1, 224, 0.3
    else:
        D, H, N, B, res, rate = 128, 2, 5, 2, 32, 0.5
    m, st = build_ref(mod, res, T, 16, D, 2, H, seed, drop_path_rate=rate)
    blk = m.transformer.resblocks[1]
    assert abs(blk.drop_path.drop_prob - rate) < 1e-6
    blk.train()
    x = randn((N, B * T, D), seed + 1).requires_grad_(True)
    g = randn((N, B * T, D), seed + 2)
    torch.manual_seed(seed + 9)
    del _drawn()[:]
    y = blk(x)
    m1, m2 = _drawn()
    assert (m1 == 0).any() and (m2 == 0).any() and not torch.equal(m1, m2), "pick another seed: masks must be non-trivial"
    params = {n: p for n, p in blk.named_parameters() if p.requires_grad}
    grads = torch.autograd.grad(y, [x] + list(params.values()), g)
    out = dict(m1=m1, m2=m2, meta=np.array([D, H, N, B, T, seed]), rate=np.float64(rate))
    if real:        # big tensors as samples (x and g are rebuilt from their seeds)
        for k, t in (("y", y), ("dx", grads[0])):
            out.update({f"{k}.{a}": b for a, b in _sample(t, 8192, seed + 3).items()})
        for (n, _), gr in zip(params.items(), grads[1:]):
            out.update({f"grad.{n}.{a}": b for a, b in _sample(gr, 2048, seed + 4).items()})
        name = f"block_real_T{T}_droppath.npz"
    else:
        out.update(x=x, g=g, y=y, dx=grads[0])
        for (n, _), gr in zip(params.items(), grads[1:]):
            out["grad." + n] = gr
        name = f"block_tiny_T{T}_droppath.npz"
    np.savez_compressed(os.path.join(HERE, name), **npify(out))


def gen_backbone_droppath(mod, T, seed):
    """Whole tiny backbone (3 layers, rates linspace(0, .5, 3)) in TRAIN mode: output, all trainable gradients and
    the 4 masks the reference drew (layers 1 and 2, two draws each)."""
    D, H, L, B = 128, 2, 3, 2
    m, st = build_ref(mod, 32, T, 16, D, L, H, seed, drop_path_rate=0.5)
    m.train()
    imgs = randn((B, 3, T, 32, 32), seed + 1)
    g = randn((B, D, T, 1, 1), seed + 2)
    torch.manual_seed(seed + 9)
    del _drawn()[:]
    y = m(imgs)
    masks = list(_drawn())
    assert len(masks) == 4 and any((k == 0).any() for k in masks)
    params = {n: p for n, p in m.named_parameters() if p.requires_grad}
    grads = torch.autograd.grad(y, list(params.values()), g)
    out = dict(imgs=imgs, g=g, y=y, meta=np.array([D, H, L, B, T, seed]), masks=torch.stack(masks))
    for (n, _), gr in zip(params.items(), grads):
        out["grad." + n] = gr
    np.savez_compressed(os.path.join(HERE, f"backbone_tiny_T{T}_droppath.npz"), **npify(out))


def gen_backbone_tiny(mod, T, seed):
    D, H, L, B = 128, 2, 2, 2
    m, st = build_ref(mod, 32, T, 16, D, L, H, seed)
    imgs = randn((B, 3, T, 32, 32), seed + 1)
    g = randn((B, D, T, 1, 1), seed + 2)
    y = m(imgs)
    params = {n: p for n, p in m.named_parameters() if p.requires_grad}
    assert sorted(params) == sorted(O.trainable_names(st))
    grads = torch.autograd.grad(y, list(params.values()), g)
    out = dict(imgs=imgs, g=g, y=y, meta=np.array([D, H, L, B, T, seed]))
    for (n, _), gr in zip(params.items(), grads):
        out["grad." + n] = gr
    # recognizer-level pins (plain torch restatement of i3d_head.py / cross_entropy_loss.py)
    C = 7
    fc_w = randn((C, D), seed + 5) * 0.1
    fc_b = randn((C,), seed + 6) * 0.1
    label = torch.tensor([3, 5][:B])
    score = O.ref_i3d_head(y.detach(), fc_w, fc_b)
    out.update(fc_w=fc_w, fc_b=fc_b, label=label, cls_score=score,
               loss_cls=O.ref_cross_entropy(score, label), pred=score.argmax(1))
    np.savez_compressed(os.path.join(HERE, f"backbone_tiny_T{T}.npz"), **npify(out))


def gen_cfg1(mod, seed):
    """BASELINE.json configs[0]: ViT-B/16, 2 frames 224^2, batch 1, fp32 CPU forward."""
    T = 2
    m, st = build_ref(mod, 224, T, 16, 768, 12, 12, seed)
    imgs = randn((1, 3, T, 224, 224), 2)
    with torch.no_grad():
        y = m(imgs)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y_bf = m(imgs).float()
    out = dict(y=y, y_autocast_bf16=y_bf, meta=np.array([768, 12, 12, 1, T, seed]))
    np.savez_compressed(os.path.join(HERE, "backbone_cfg1.npz"), **npify(out))


def main():
    torch.set_num_threads(8)
    mod = load_reference()
    if "--aim-only" in sys.argv:            # stock-AIM variant (vitclip_aim.py), round 2
        aim = load_reference_aim()
        gen_aim(aim, 2, 1100, False)
        gen_aim(aim, 4, 1200, True)
        return
    if "--droppath-only" in sys.argv:       # round 2 additions; the round-1 fixtures stay byte-identical
        gen_block_droppath(mod, 2, 700)
        gen_block_droppath(mod, 4, 800)
        gen_block_droppath(mod, 2, 900, real=True)
        gen_backbone_droppath(mod, 2, 1000)
        return
    gen_block(mod, 2, 100)
    gen_block(mod, 4, 200)
    gen_block_real(mod, 300)
    gen_backbone_tiny(mod, 2, 400)
    gen_backbone_tiny(mod, 4, 500)
    gen_cfg1(mod, 600)
    gen_block_droppath(mod, 2, 700)
    gen_block_droppath(mod, 4, 800)
    gen_block_droppath(mod, 2, 900, real=True)
    gen_backbone_droppath(mod, 2, 1000)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
