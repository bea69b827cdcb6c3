import torch, time
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 100864
a = torch.empty((M, 2304), dtype=torch.bfloat16, device="cuda")
b = torch.randn((M, 768), device="cuda"); c = torch.empty_like(b); d = torch.randn_like(b)
ms = t(lambda: a.fill_(1.0)); print(f"fill bf16 465MB: {ms:.3f} ms  {a.numel()*2/ms/1e9:.2f} TB/s")
ms = t(lambda: c.fill_(1.0)); print(f"fill f32 310MB: {ms:.3f} ms  {c.numel()*4/ms/1e9:.2f} TB/s")
ms = t(lambda: torch.add(b, d, out=c)); print(f"add f32 (2 reads 1 write, 930MB): {ms:.3f} ms  {3*c.numel()*4/ms/1e9:.2f} TB/s")
ms = t(lambda: c.copy_(b)); print(f"copy f32 (620MB): {ms:.3f} ms  {2*c.numel()*4/ms/1e9:.2f} TB/s")
big = torch.empty((M, 3072), dtype=torch.bfloat16, device="cuda"); big2 = torch.empty_like(big)
ms = t(lambda: big2.copy_(big)); print(f"copy bf16 (1240MB): {ms:.3f} ms  {2*big.numel()*2/ms/1e9:.2f} TB/s")
