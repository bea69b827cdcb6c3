import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from aim_amd.dist import build_optimizer
dev = torch.device("cuda", 0)
model = bench.build_model(8, dev)
opt = build_optimizer(model, dict(type='AdamW', lr=3e-4, weight_decay=0.05))
imgs = torch.randn((64, 1, 3, 8, 224, 224)).to(dev); label = torch.randint(0, 400, (64, 1)).to(dev)
def step():
    opt.zero_grad(); l = model(imgs, label, return_loss=True)["loss_cls"]; l.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue time/step {1e3*(t1-t0)/5:.1f} ms ; wall/step {1e3*(t2-t0)/5:.1f} ms")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5): step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
