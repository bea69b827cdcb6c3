"""Where do the device-to-device copies of a step come from?  torch.profiler with stacks over one inference / train step;
prints every aten op that ends in a copy kernel with its Python call site.  usage: python tools/prof_copies.py [inf|sec|main]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "inf"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if which == "inf":
    import aim_amd
    a = bench.ARCH["L14"]
    cfg = dict(type='Recognizer3D', backbone=dict(type='ViT_CLIP', input_resolution=224, num_frames=32, drop_path_rate=0.2, adapter_scale=0.5,
                                                  pretrained=None, **a),
               cls_head=dict(type='I3DHead', in_channels=a["width"], num_classes=400, spatial_type='avg', dropout_ratio=0.5),
               test_cfg=dict(average_clips='prob'))
    model = aim_amd.build_model(cfg).to(dev).eval()
    model.backbone.set_inference_precision(os.environ.get("PREC", "fp8"))
    imgs = torch.randn((4, 3, 3, 32, 224, 224), device=dev)

    def step():
        with torch.no_grad():
            model._do_test(imgs)
else:
    from aim_amd.dist import build_optimizer
    model = bench.build_model(16 if which == "sec" else 8, dev, "L14" if which == "sec" else "B16")
    opt = build_optimizer(model, dict(type='AdamW', lr=3e-4, weight_decay=0.05))
    B = 8 if which == "sec" else 16
    T = 16 if which == "sec" else 8
    imgs = torch.randn((B, 1, 3, T, 224, 224), device=dev)
    label = torch.randint(0, 400, (B, 1), device=dev)

    def step():
        opt.zero_grad()
        model(imgs, label, return_loss=True)["loss_cls"].backward()
        opt.step()

for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name in ("aten::copy_", "aten::cat", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::zeros"):
        st = [s for s in (ev.stack or []) if "adapt-image-models_amd" in s or "bench.py" in s or "prof_copies" in s]
        key = (ev.name, st[0] if st else "?", str(ev.input_shapes)[:80])
        d = rows.setdefault(key, [0, 0.0])
        d[0] += 1
        d[1] += ev.device_time_total
for k, (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:4d} x {us / 1e3:8.3f} ms  {k[0]:18s} {k[1]}  {k[2]}")
