import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
M, N, K = 1024, 256, 64
a = torch.zeros((M, K), device="cuda")
a[:, 0] = (torch.arange(M, device="cuda") % 256).float()
w = torch.zeros((N, K), device="cuda"); w[:, 0] = 1.0
out = torch.full((M, N), -1.0, dtype=torch.bfloat16, device="cuda")
ops.gemm(a.to(torch.bfloat16), w.to(torch.bfloat16), ops.EPI_BF16, out)
o = out.float().cpu()
print("col0 rows 0..47:", o[:48, 0].int().tolist())
print("row 8 cols 0..40:", o[8, :40].int().tolist())
print("row 300 cols 60..70:", o[300, 60:70].int().tolist())
# column pattern: W[n][0] = n, A[m][0] = 1
a2 = torch.zeros((M, K), device="cuda"); a2[:, 0] = 1.0
w2 = torch.zeros((N, K), device="cuda"); w2[:, 0] = torch.arange(N, device="cuda").float()
ops.gemm(a2.to(torch.bfloat16), w2.to(torch.bfloat16), ops.EPI_BF16, out)
o = out.float().cpu()
print("row 0 cols 0..40:", o[0, :40].int().tolist())
print("row 9 cols 0..40:", o[9, :40].int().tolist())
