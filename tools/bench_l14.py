"""BASELINE configs[3] shape on ONE GPU: ViT-L/14 + AIM, 16 frames 224^2, 32 clips per GPU (256 / 8), fwd + bwd + AdamW."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aim_amd
from aim_amd.dist import build_optimizer
dev = torch.device("cuda", 0)
frames, B = int(os.environ.get("FRAMES", 16)), int(os.environ.get("B", 32))
cfg = dict(type='Recognizer3D',
           backbone=dict(type='ViT_CLIP', input_resolution=224, patch_size=14, num_frames=frames, width=1024, layers=24,
                         heads=16, drop_path_rate=0.2, adapter_scale=0.5, pretrained=None),
           cls_head=dict(type='I3DHead', in_channels=1024, num_classes=400, spatial_type='avg', dropout_ratio=0.5),
           test_cfg=dict(average_clips='prob'))
torch.manual_seed(0)
model = aim_amd.build_model(cfg)
with torch.no_grad():
    for n, p in model.named_parameters():
        if "D_fc2" in n:
            p.normal_(0, 0.02)
model = model.to(dev).train()
opt = build_optimizer(model, dict(type='AdamW', lr=3e-4, weight_decay=0.05))
x = torch.randn((B, 1, 3, frames, 224, 224), device=dev)
y = torch.randint(0, 400, (B, 1), device=dev)
def step():
    opt.zero_grad(); loss = model(x, y, return_loss=True)["loss_cls"]; loss.backward(); opt.step(); return loss
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 4
for _ in range(n): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
# ViT-L/14: per token-layer 12*D^2*... use measured-time only; report clips/s and peak memory
print(f"ViT-L/14 T={frames} B={B}: {B / dt:.1f} clips/s, {dt * 1e3:.1f} ms/step, loss {float(loss):.4f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
