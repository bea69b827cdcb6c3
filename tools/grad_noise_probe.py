"""How far the bf16 product's gradients are from the fp32 mode's on the GPU, per tensor (relative L2), for the build's
switchable rounding choices -- run once per environment (AIM_AUX_GRAD=0|1 is read at import):
    python tools/grad_noise_probe.py [vitb16|vitl14] [frames] [clips]
Prints worst / median / by-adapter medians and the forward distance."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aim_amd  # noqa: E402
from oracle import vit_clip_oracle as O  # noqa: E402  (synthetic weights only)

geo = sys.argv[1] if len(sys.argv) > 1 else "vitb16"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
patch, D, L, H = (16, 768, 12, 12) if geo == "vitb16" else (14, 1024, 24, 16)
m = aim_amd.ViT_CLIP(224, T, patch, D, L, H, 0.0)
m.init_weights()
st = O.synth_state_dict(O.backbone_param_shapes(224, T, patch, D, L), seed=77)
m.load_state_dict(st, strict=True)
m = m.cuda().eval()
gen = torch.Generator().manual_seed(5)
imgs = torch.randn((B, 3, T, 224, 224), generator=gen).cuda()
g = torch.randn((B, D, T, 1, 1), generator=gen).cuda()


def grads(prec):
    m.set_precision(prec)
    for p in m.parameters():
        p.grad = None
    y = m(imgs)
    y.backward(g)
    return y.detach(), {n: p.grad.double() for n, p in m.named_parameters() if p.requires_grad}


y32, g32 = grads('fp32')
y16, g16 = grads('bf16')
rel = {n: ((g16[n] - g32[n]).norm() / g32[n].norm()).item() for n in g32}
worst = max(rel, key=rel.get)
print(f"{geo} T={T} B={B} AIM_AUX_GRAD={os.environ.get('AIM_AUX_GRAD', '1')}: forward rel L2 "
      f"{((y16.double() - y32.double()).norm() / y32.double().norm()).item():.3e}")
print(f"  gradients: worst {rel[worst]:.3e} ({worst}), median {np.median(list(rel.values())):.3e}")
for a in ("MLP_Adapter", "S_Adapter", "T_Adapter", "ln_post", "temporal"):
    v = [r for n, r in rel.items() if a in n]
    print(f"  {a:12s} median {np.median(v):.3e} max {max(v):.3e}")
# by layer (MLP_Adapter.D_fc1.weight: a sum over all token rows): the residual-stream gradient is rounded to bf16 twice per
# block on its way down, so a growth towards layer 0 is that rounding; a flat profile says it is not what bounds the error
print("  MLP_Adapter.D_fc1.weight by layer:", " ".join(
    f"{rel[f'transformer.resblocks.{i}.MLP_Adapter.D_fc1.weight']:.2e}" for i in range(L)))
print("  T_Adapter.D_fc1.weight by layer:  ", " ".join(
    f"{rel[f'transformer.resblocks.{i}.T_Adapter.D_fc1.weight']:.2e}" for i in range(L)))
