"""The step runs on three HIP streams (main / class-token side / detached weight gradients).  None of its kernels uses
atomics, so two identical short trainings must end in BIT-identical parameters and losses -- at the real ViT-B/16 shape,
where the streams really overlap.  A cross-stream race (a buffer reused before its reader ran, a missing join) shows
up here as a difference."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _train(steps=3, B=64, frames=8):      # BASELINE configs[1]: 64 clips per GPU
    import bench
    from aim_amd.dist import build_optimizer
    dev = torch.device("cuda", 0)
    torch.manual_seed(123)
    model = bench.build_model(frames, dev)
    opt = build_optimizer(model, dict(type='AdamW', lr=3e-4, betas=(0.9, 0.999), weight_decay=0.05))
    g = torch.Generator(device="cpu").manual_seed(7)
    imgs = torch.randn((B, 1, 3, frames, 224, 224), generator=g).to(dev)
    label = torch.randint(0, 400, (B, 1), generator=g).to(dev)
    losses = []
    torch.manual_seed(99)            # DropPath masks / dropout draw from the default generators
    torch.cuda.manual_seed(99)
    for _ in range(steps):
        opt.zero_grad()
        loss = model(imgs, label, return_loss=True)["loss_cls"]
        loss.backward()
        opt.step()
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), opt.flat_p.detach().cpu().clone(), opt.flat_g.detach().cpu().clone()


def test_two_identical_trainings_are_bit_identical():
    l1, p1, g1 = _train()
    l2, p2, g2 = _train()
    assert torch.isfinite(l1).all() and torch.isfinite(p1).all()
    assert torch.equal(l1, l2), (l1, l2)
    assert torch.equal(g1, g2), (g1 - g2).abs().max()
    assert torch.equal(p1, p2), (p1 - p2).abs().max()
