"""The drop-in claim, pinned: every ``configs/recognition/vit/vitclip_{base,large}_*.py`` of the reference goes through
``aim_amd.Config.fromfile`` (``_base_`` inheritance included) and ``aim_amd.build_model`` UNCHANGED, and the model that
comes out has the reference's architecture numbers, freeze policy and state_dict keys.

The config files are read where they lie under ``/root/reference`` (nothing is copied into this repository); on a box
without the reference (the GPU box) the whole module is skipped.  Two configs cannot build, and must fail the way the
reference itself does:

* ``vitclip_base_hmdb51.py``: ``pretrained='openaiclip'`` -> ``clip.load`` (vit_clip.py:369-372); the ``clip`` package
  and its downloaded weights are absent here, so ``init_weights`` raises the documented ``RuntimeError``;
* ``vitclip_base_sthv2.py`` passes ``num_tadapter=2``, a keyword the reference's own ``ViT_CLIP.__init__``
  (vit_clip.py:330-331) does not take: ``TypeError`` there and here.
"""
import glob
import os

import pytest
import torch

REF_CFG = "/root/reference/configs/recognition/vit"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference tree not present on this box")

FILES = sorted(glob.glob(os.path.join(REF_CFG, "vitclip_*.py")))
ARCH = {"base": dict(width=768, layers=12, heads=12, patch_size=16, tokens=197),
        "large": dict(width=1024, layers=24, heads=16, patch_size=14, tokens=257)}
KNOWN_FAIL = {"vitclip_base_hmdb51.py": (RuntimeError, "clip"), "vitclip_base_sthv2.py": (TypeError, "num_tadapter")}


def test_the_nine_configs_are_there():
    assert len(FILES) == 9 and set(KNOWN_FAIL) <= {os.path.basename(f) for f in FILES}


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_reference_config_builds_unchanged(path):
    import aim_amd
    name = os.path.basename(path)
    cfg = aim_amd.Config.fromfile(path)
    m = cfg.model
    assert m.type == "Recognizer3D" and m.backbone.type == "ViT_CLIP" and m.cls_head.type == "I3DHead"
    assert m.test_cfg.average_clips == "prob"                                  # _base_/models/vitclip_base.py:19
    arch = ARCH["large" if "_large_" in name else "base"]
    assert (m.backbone.width, m.backbone.layers, m.backbone.heads, m.backbone.patch_size) == \
        (arch["width"], arch["layers"], arch["heads"], arch["patch_size"])
    # the training-side keys the hot path's callers read (optimizer groups, the accumulation hook, the fused normalise hook)
    assert cfg.optimizer.type == "AdamW" and "custom_keys" in cfg.optimizer.paramwise_cfg
    assert cfg.optimizer_config.type == "DistOptimizerHook" and cfg.optimizer_config.update_interval >= 1
    if name in KNOWN_FAIL:
        exc, what = KNOWN_FAIL[name]
        with pytest.raises(exc, match=what):
            aim_amd.build_model(m)
        return
    torch.manual_seed(0)
    model = aim_amd.build_model(m)
    bb = model.backbone
    assert isinstance(bb, aim_amd.ViT_CLIP) and bb.num_frames == m.backbone.num_frames
    assert bb.positional_embedding.shape == (arch["tokens"], arch["width"])
    assert bb.temporal_embedding.shape == (1, m.backbone.num_frames, arch["width"])
    assert model.cls_head.fc_cls.weight.shape == (m.cls_head.num_classes, arch["width"])
    if "max_testing_views" in m.test_cfg:
        assert model.max_testing_views == m.test_cfg.max_testing_views
    # freeze policy (vit_clip.py:413-415): 12 adapter tensors per layer + temporal_embedding + ln_post.{w,b} + the head
    train = [n for n, p in model.named_parameters() if p.requires_grad]
    assert len(train) == 12 * arch["layers"] + 3 + 2
    assert all(any(k in n for k in ("Adapter", "ln_post", "temporal_embedding", "cls_head")) for n in train)
    assert all(float(p.abs().max()) == 0 for n, p in model.named_parameters() if "D_fc2" in n)      # :386-411
    # parameter groups exactly as mmcv's DefaultOptimizerConstructor would cut them from this config (decay_mult = 0 keys)
    wd = cfg.optimizer.weight_decay
    for n, p in bb.named_parameters():
        if not p.requires_grad:
            continue
        mult = [v.get("decay_mult", 1.0) for k, v in cfg.optimizer.paramwise_cfg.custom_keys.items() if k in "backbone." + n]
        assert (wd * mult[0] if mult else wd) in (0.0, wd)
    # GPUNormalize pre-hooks of the configs that use them register on the backbone (module_hooks.py:8-32)
    hooks = cfg.get("module_hooks")
    if hooks:
        handles = aim_amd.register_module_hooks(model, [dict(h) for h in hooks])
        assert len(handles) == len(hooks) and len(bb._forward_pre_hooks) == len(hooks)
    # the accumulation hook takes the config's keywords as they are
    oc = dict(cfg.optimizer_config)
    assert oc.pop("type") == "DistOptimizerHook"
    assert aim_amd.DistOptimizerHook(**oc).update_interval == cfg.optimizer_config.update_interval
