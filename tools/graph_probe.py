"""Is the reference's own evaluation protocol (ONE sample x 3 views per call, recognizer3d.py:38-60) bound by the host?
Times `Recognizer3D._do_test` on ViT-L/14, 32 frames: eager (whole step and the host's enqueue time alone) against a captured
HIP graph of the same call (torch.cuda.graph: every kernel of this package is launched on torch's current stream, side
streams are forked and joined by events, all buffers come from torch's allocator -- nothing in the path synchronises).
    python tools/graph_probe.py [samples] [fp8|bf16]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aim_amd  # noqa: E402
import bench  # noqa: E402

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prec = sys.argv[2] if len(sys.argv) > 2 else "fp8"
dev = torch.device("cuda", 0)
a = bench.ARCH["L14"]
cfg = dict(type='Recognizer3D',
           backbone=dict(type='ViT_CLIP', input_resolution=224, num_frames=32, drop_path_rate=0.2, adapter_scale=0.5, pretrained=None, **a),
           cls_head=dict(type='I3DHead', in_channels=a["width"], num_classes=400, spatial_type='avg', dropout_ratio=0.5),
           test_cfg=dict(average_clips='prob'))
torch.manual_seed(0)
model = aim_amd.build_model(cfg)
with torch.no_grad():
    for n, p in model.named_parameters():
        if "D_fc2" in n:
            p.normal_(0, 0.02)
model = model.to(dev).eval()
model.backbone.set_inference_precision(prec)
imgs = torch.randn((samples, 3, 3, 32, 224, 224), generator=torch.Generator().manual_seed(1)).to(dev)
steps = 20
with torch.no_grad():
    for _ in range(3):
        ref = model._do_test(imgs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model._do_test(imgs)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{prec} {samples} sample(s) x 3 views eager: {t_all / steps * 1e3:.2f} ms per call ({samples * 3 * steps / t_all:.1f} views/s); "
          f"host enqueue alone {t_host / steps * 1e3:.2f} ms per call")
    g = torch.cuda.CUDAGraph()
    static_in = imgs.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model._do_test(static_in)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        static_out = model._do_test(static_in)
    torch.cuda.synchronize()
    static_in.copy_(imgs)
    g.replay()
    torch.cuda.synchronize()
    print("graph output equals eager:", bool(torch.equal(static_out, ref)), float((static_out - ref).abs().max()))
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    t_g = time.perf_counter() - t0
    print(f"captured graph: {t_g / steps * 1e3:.2f} ms per call ({samples * 3 * steps / t_g:.1f} views/s)")
if os.environ.get("PROFILE"):
    import cProfile
    import pstats
    with torch.no_grad():
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(10):
            model._do_test(imgs)
        pr.disable()
        torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
