"""Which gradients differ between two identical forward+backward passes (same parameters, same RNG state)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
model = bench.build_model(8, dev)
B = int(os.environ.get("B", 8))
g = torch.Generator(device="cpu").manual_seed(7)
imgs = torch.randn((B, 1, 3, 8, 224, 224), generator=g).to(dev)
label = torch.randint(0, 400, (B, 1), generator=g).to(dev)
def run():
    torch.manual_seed(99); torch.cuda.manual_seed(99)
    model.zero_grad(set_to_none=True)
    loss = model(imgs, label, return_loss=True)["loss_cls"]
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
l1, g1 = run()
for rep in range(3):
    l2, g2 = run()
    bad = [(n, (g1[n] - g2[n]).abs().max().item(), g1[n].abs().max().item()) for n in g1 if not torch.equal(g1[n], g2[n])]
    print(f"rep {rep}: loss equal {torch.equal(l1, l2)} ({l1.item():.6f} vs {l2.item():.6f}); {len(bad)} of {len(g1)} gradient tensors differ")
    for n, d, m in bad[:12]:
        print(f"   {n:60s} max diff {d:.3e} (max |g| {m:.3e})")
