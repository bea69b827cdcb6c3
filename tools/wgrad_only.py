import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
M = 100864
g = torch.randn((M, 768), device="cuda").to(torch.bfloat16); a = torch.randn((M, 192), device="cuda").to(torch.bfloat16)
dw = torch.zeros((768, 192), device="cuda")
for _ in range(3): ops.wgrad(g, a, dw)
torch.cuda.synchronize()
