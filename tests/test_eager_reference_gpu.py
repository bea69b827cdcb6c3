"""Reference point, not a parity test: the oracle's literal PyTorch restatement of the reference (eager ops, the way the
reference itself runs under torch.autocast) on the SAME GPU, ViT-B/16 8x224^2, forward + backward + head, a few clips.
Prints clips/s; asserts only that it ran and that the product path is faster."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_eager_torch_port_throughput(capsys):
    from oracle import vit_clip_oracle as O
    dev = torch.device("cuda", 0)
    frames, B = 8, 32
    st = O.synth_state_dict(O.backbone_param_shapes(224, frames, 16, 768, 12), seed=0)
    st = {k: v.to(dev) for k, v in st.items()}
    names = O.trainable_names(st)
    for n in names:
        st[n].requires_grad_(True)
    fc_w = (torch.randn(400, 768, device=dev) * 0.01).requires_grad_(True)
    fc_b = torch.zeros(400, device=dev, requires_grad=True)
    imgs = torch.randn(B, 3, frames, 224, 224, device=dev)
    label = torch.randint(0, 400, (B,), device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = O.ref_backbone(imgs, st, 12, frames)
            loss = O.ref_cross_entropy(O.ref_i3d_head(y.float(), fc_w, fc_b), label)
        torch.autograd.grad(loss, [st[n] for n in names] + [fc_w, fc_b])

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 4
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    eager = B * n / (time.perf_counter() - t0)

    import aim_amd
    import bench
    model = bench.build_model(frames, dev)
    x = torch.randn((B, 1, 3, frames, 224, 224), device=dev)
    lab = torch.randint(0, 400, (B, 1), device=dev)

    def pstep():
        model.zero_grad(set_to_none=True)
        model(x, lab, return_loss=True)["loss_cls"].backward()

    for _ in range(2):
        pstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        pstep()
    torch.cuda.synchronize()
    ours = B * n / (time.perf_counter() - t0)
    with capsys.disabled():
        print(f"\n[reference point] eager PyTorch-ROCm port (oracle ref_*, autocast bf16), {B} clips/step: {eager:.1f} clips/s;"
              f" this framework, same {B} clips/step: {ours:.1f} clips/s ({ours / eager:.1f}x)")
    assert ours > eager
