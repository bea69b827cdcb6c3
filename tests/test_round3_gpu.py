"""Round-3 behaviour pins on a real MI355X (each cites what it answers):

* ``checkpoint=True`` (vit_clip.py:316-320) recomputes each block's context in the backward: bit-identical losses,
  gradients and parameters to ``checkpoint=False`` with DropPath ON, at a fraction of the activation memory;
* fp8 operand cache vs ``FlatAdamW`` (raw-pointer updates): an fp8 eval after an optimizer step must see the updated
  adapter weights (train -> fp8 eval -> train -> fp8 eval in one process);
* ``aim_ce_topk`` with labels outside [0, C): ignored exactly like ``F.cross_entropy``'s ignore_index = -100.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _recognizer(variant, checkpoint, T=4, L=3, res=64, D=128, H=2, drop=0.3):
    import aim_amd
    bb = dict(type=variant, input_resolution=res, patch_size=16, num_frames=T, width=D, layers=L, heads=H,
              drop_path_rate=drop, adapter_scale=0.5, pretrained=None)
    if variant == "ViT_CLIP":
        bb["checkpoint"] = checkpoint
    cfg = dict(type='Recognizer3D', backbone=bb,
               cls_head=dict(type='I3DHead', in_channels=D, num_classes=7, dropout_ratio=0.0),
               test_cfg=dict(average_clips='prob'))
    torch.manual_seed(11)
    model = aim_amd.build_model(cfg)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "D_fc2" in n or "temporal_embedding" in n:
                p.normal_(0, 0.02)
    if variant == "AIM":
        model.backbone.checkpoint = checkpoint        # (the stock-AIM ctor has no such keyword: vitclip_aim.py:356-358)
    return model.to(DEV).train()


def _train(variant, checkpoint, steps=2, B=6):
    from aim_amd.dist import build_optimizer
    model = _recognizer(variant, checkpoint)
    opt = build_optimizer(model, dict(type='AdamW', lr=1e-2, weight_decay=0.05))
    g = torch.Generator().manual_seed(5)
    imgs = torch.randn((B, 1, 3, 4, 64, 64), generator=g).to(DEV)
    label = torch.randint(0, 7, (B, 1), generator=g).to(DEV)
    torch.manual_seed(77); torch.cuda.manual_seed(77)          # the DropPath draws of both runs
    losses, grads = [], []
    for _ in range(steps):
        opt.zero_grad()
        loss = model(imgs, label, return_loss=True)["loss_cls"]
        loss.backward()
        grads.append(opt.flat_g.detach().clone())
        opt.step()
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), [g_.cpu() for g_ in grads], opt.flat_p.detach().cpu().clone()


@pytest.mark.parametrize("variant", ["ViT_CLIP", "AIM"])
def test_checkpoint_recompute_is_bitwise_identical(variant):
    l0, g0, p0 = _train(variant, False)
    l1, g1, p1 = _train(variant, True)
    assert torch.isfinite(l0).all() and float(g0[0].abs().max()) > 0
    assert torch.equal(l0, l1), (l0, l1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b), (a - b).abs().max()
    assert torch.equal(p0, p1)


def test_checkpoint_keeps_one_block_context():
    """ViT-B/16 width, 8 clips x 8 frames, 8 layers: the no-checkpoint run keeps ~0.26 GiB of context per layer (plus what
    the detached weight-gradient stream still holds), the checkpointed one a block input per layer plus ONE block's
    context and backward transients at a time."""
    import aim_amd
    peaks = {}
    for ck in (False, True):
        torch.manual_seed(0)
        m = aim_amd.ViT_CLIP(224, 8, 16, 768, 8, 12, 0.1, checkpoint=ck)
        m.init_weights()
        m = m.to(DEV).train()
        x = torch.randn(8, 3, 8, 224, 224, device=DEV)
        m(x[:1]).sum().backward()                 # operand staging outside the measured region
        torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        y = m(x)
        y.float().square().mean().backward()
        torch.cuda.synchronize()
        peaks[ck] = torch.cuda.max_memory_allocated() - base
        del m, x, y
        torch.cuda.empty_cache()
    assert peaks[True] < 0.5 * peaks[False], peaks


def test_fp8_eval_follows_the_optimizer():
    """`_fp8_operands()` was keyed on (data_ptr, _version), which FlatAdamW never changes."""
    import aim_amd
    from aim_amd.dist import build_optimizer
    cfg = dict(type='Recognizer3D',
               backbone=dict(type='ViT_CLIP', input_resolution=224, patch_size=16, num_frames=2, width=768, layers=2, heads=12,
                             drop_path_rate=0.0, adapter_scale=0.5, pretrained=None),
               cls_head=dict(type='I3DHead', in_channels=768, num_classes=9, dropout_ratio=0.0),
               test_cfg=dict(average_clips='prob'))
    torch.manual_seed(3)
    model = aim_amd.build_model(cfg).to(DEV)          # D_fc2 = 0: the first fp8 eval quantises a ZERO MLP_Adapter
    opt = build_optimizer(model, dict(type='AdamW', lr=5e-2, weight_decay=0.0))
    g = torch.Generator().manual_seed(4)
    imgs = torch.randn((4, 1, 3, 2, 224, 224), generator=g).to(DEV)
    label = torch.randint(0, 9, (4, 1), generator=g).to(DEV)
    bb = model.backbone

    def evals():
        model.eval()
        with torch.no_grad():
            bb.set_inference_precision('bf16')
            y16 = bb(imgs[:, 0])
            bb.set_inference_precision('fp8')
            y8 = bb(imgs[:, 0])
        model.train()
        return y16.float(), y8.float()

    a16, a8 = evals()
    d0 = _rel(a8, a16)
    for _ in range(3):                                 # lr 5e-2: the adapters move a lot
        opt.zero_grad()
        model(imgs, label, return_loss=True)["loss_cls"].backward()
        opt.step()
    b16, b8 = evals()
    moved = _rel(b16, a16)
    assert moved > 5 * d0, (moved, d0)                 # the update is far outside the fp8-vs-bf16 distance ...
    assert _rel(b8, b16) < 3 * d0 + 0.05 * moved, (_rel(b8, b16), d0, moved)      # ... and the fp8 path followed it
    assert _rel(b8, a8) > 0.5 * moved


@pytest.mark.parametrize("B,C", [(8, 11), (64, 400)])
def test_ce_topk_ignores_out_of_range_labels(B, C):
    from aim_amd import ops
    g = torch.Generator().manual_seed(B + C)
    score = torch.randn((B, C), generator=g).to(DEV)
    label = torch.randint(0, C, (B,), generator=g)
    label[1], label[B - 2] = -100, C                    # torch's ignore_index, and one past the end
    lab_t = label.clone()
    lab_t[B - 2] = -100                                 # (torch device-asserts on C itself; here it is ignored too)
    sr = score.clone().requires_grad_(True)
    ref = F.cross_entropy(sr, lab_t.to(DEV), ignore_index=-100)
    ref.backward()
    out3, dscore = ops.ce_topk(score, label.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(out3[0]) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert _rel(dscore, sr.grad) < 1e-5
    assert float(dscore[1].abs().max()) == 0.0 and float(dscore[B - 2].abs().max()) == 0.0
    valid = (lab_t >= 0)
    top1 = (score.cpu().argmax(1) == lab_t).float()[valid].sum() / B       # an ignored sample counts as a miss
    assert abs(float(out3[1]) - float(top1)) < 1e-6
    # all labels ignored: torch returns nan for the mean over zero samples; so does the kernel
    out3b, dsb = ops.ce_topk(score, torch.full((B,), -100, dtype=torch.int64, device=DEV))
    assert out3b[0].isnan() and float(dsb.abs().max()) == 0.0
