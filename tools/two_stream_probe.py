"""Do two independent inference batches on two HIP streams fill each other's tile-round tails?  ViT-L/14, 32 frames, fp8:
one stream with 12 views per call against two threads (own streams) with 6 views per call each.
    python tools/two_stream_probe.py [fp8|bf16]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aim_amd  # noqa: E402
import bench  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp8"
dev = torch.device("cuda", 0)
a = bench.ARCH["L14"]
cfg = dict(type='Recognizer3D',
           backbone=dict(type='ViT_CLIP', input_resolution=224, num_frames=32, drop_path_rate=0.2, adapter_scale=0.5, pretrained=None, **a),
           cls_head=dict(type='I3DHead', in_channels=a["width"], num_classes=400, spatial_type='avg', dropout_ratio=0.5),
           test_cfg=dict(average_clips='prob'))
torch.manual_seed(0)
model = aim_amd.build_model(cfg).to(dev).eval()
model.backbone.set_inference_precision(prec)
imgs = torch.randn((4, 3, 3, 32, 224, 224), generator=torch.Generator().manual_seed(1)).to(dev)
steps = 8


def run(x, n, stream=None):
    with torch.no_grad():
        if stream is None:
            for _ in range(n):
                model._do_test(x)
        else:
            with torch.cuda.stream(stream):
                for _ in range(n):
                    model._do_test(x)


run(imgs, 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(imgs, steps)
torch.cuda.synchronize()
t1 = time.perf_counter() - t0
print(f"{prec} one stream, 12 views per call: {12 * steps / t1:.1f} views/s ({t1 / steps * 1e3:.1f} ms per 12 views)")
halves = [imgs[:2].contiguous(), imgs[2:].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for h, s in zip(halves, streams):
    s.wait_stream(torch.cuda.current_stream())
    run(h, 1, s)
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(h, steps, s)) for h, s in zip(halves, streams)]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
t2 = time.perf_counter() - t0
print(f"{prec} two streams, 6 views per call each: {12 * steps / t2:.1f} views/s ({t2 / steps * 1e3:.1f} ms per 12 views)")
run(halves[0], 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(halves[0], steps)
torch.cuda.synchronize()
t3 = time.perf_counter() - t0
print(f"{prec} one stream, 6 views per call: {6 * steps / t3:.1f} views/s")
