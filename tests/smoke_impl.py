"""smoke(): one tiny forward+backward of the HIP backbone on cuda:0, checked against the CPU oracle."""
import torch

from oracle import vit_clip_oracle as O


def run_smoke():
    import aim_amd
    T, D, L, H = 2, 128, 2, 2
    m = aim_amd.ViT_CLIP(32, T, 16, D, L, H, 0.0)
    m.init_weights()
    st = O.synth_state_dict(O.backbone_param_shapes(32, T, 16, D, L), seed=11)
    m.load_state_dict(st, strict=True)
    m = m.to("cuda:0").eval()
    imgs = torch.randn((2, 3, T, 32, 32), generator=torch.Generator().manual_seed(5))
    g = torch.randn((2, D, T, 1, 1), generator=torch.Generator().manual_seed(6))
    y = m(imgs.to("cuda:0"))
    y.backward(g.to("cuda:0"))
    names = O.trainable_names(st)
    for n in names:
        st[n] = st[n].detach().requires_grad_(True)
    y_ref = O.emu_backbone(imgs, st, H, rnd=O.BF16)
    grads = torch.autograd.grad(y_ref, [st[n] for n in names], g)
    # same bars as tests/test_backbone_gpu.py: relative L2 <= 3e-3, max-abs within 3 bf16 ulps of O(1) outputs, gradients <= 2.5e-2
    err = (y.detach().cpu() - y_ref.detach()).abs().max().item()
    rel = ((y.detach().cpu() - y_ref.detach()).norm() / y_ref.detach().norm()).item()
    assert rel < 3e-3 and err < 1.2e-2, f"smoke forward mismatch: rel {rel} max {err}"
    got = dict(m.named_parameters())
    for n, gr in zip(names, grads):
        e = ((got[n].grad.cpu() - gr).norm() / (gr.norm() + 1e-30)).item()
        assert e < 2.5e-2, f"smoke grad mismatch {n}: {e}"
    # the reference-precision mode (csrc/fp32.hip) against the oracle's literal fp32 restatement of vit_clip.py:433-458
    # (itself pinned to the reference's fixtures): north_star's 1e-5 bar, max |a - b| / max |b|
    with torch.no_grad():
        y32 = m.set_precision('fp32')(imgs.to("cuda:0")).cpu()
        ref32 = O.ref_backbone(imgs, {k: v.detach() for k, v in st.items()}, H, T)
    e32 = ((y32 - ref32).abs().max() / ref32.abs().max()).item()
    assert e32 <= 1e-5, f"smoke fp32 mismatch: {e32}"
    # ... and its backward (the same hand-written backward on fp32 kernels) against the restatement's autograd, per tensor
    for p in m.parameters():
        p.grad = None
    m(imgs.to("cuda:0")).backward(g.to("cuda:0"))
    m.set_precision('bf16')
    st32 = {k: v.detach().clone().requires_grad_(k in names) for k, v in st.items()}
    g32 = torch.autograd.grad(O.ref_backbone(imgs, st32, H, T), [st32[n] for n in names], g)
    eg = max(((got[n].grad.cpu() - gr).abs().max() / gr.abs().max()).item() for n, gr in zip(names, g32))
    assert eg <= 1e-5, f"smoke fp32 gradient mismatch: {eg}"
    print(f"smoke ok: bf16 fwd rel err {rel:.2e}, max err {err:.2e}; fp32 mode max-rel err {e32:.2e}, gradients {eg:.2e}")
