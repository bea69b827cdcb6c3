"""The thin-tail peel of the large GEMM (csrc/gemm.hip, AIM_GEMM_PEEL): a launch whose tile count is a hair over whole rounds of
the persistent 256 x 256 kernel is cut into the whole-round rows and a low-latency launch for the last row tiles.  Same K
order, same epilogue arithmetic: the outputs must be BIT-IDENTICAL to the un-peeled launch (run in a child process with
AIM_GEMM_PEEL=0), for the linear and the residual epilogues with every per-frame / per-token factor in use."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from aim_amd import ops
torch.manual_seed(0)
out = {}
cus = torch.cuda.get_device_properties(0).multi_processor_count
for tag, N, K, ntok in (("n1024_k1024", 1024, 1024, 257), ("n1024_k4352", 1024, 4352, 257), ("n768_k768", 768, 768, 197)):
    tn = (N + 255) // 256
    rows = (2 * cus // tn) * 256 + 300            # two whole rounds + two thin row tiles (one of them ragged)
    nfr = (rows + ntok - 1) // ntok
    a = torch.randn((rows, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    bias, af, at = torch.randn(N, device="cuda"), torch.rand(nfr, device="cuda") + 0.5, torch.rand(ntok, device="cuda") + 0.5
    vec, bt, resid = torch.randn((nfr, N), device="cuda"), torch.rand(ntok, device="cuda"), torch.randn((rows, N), device="cuda")
    y16 = torch.empty((rows, N), device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, w, ops.EPI_BF16, y16, bias=bias, af=af, at=at, ntok=ntok)
    y32 = torch.empty((rows, N), device="cuda")
    ops.gemm(a, w, ops.EPI_F32, y32, bias=bias, resid=resid, af=af, vec=vec, bt=bt, ntok=ntok)
    out[tag + ".bf16"], out[tag + ".f32"], out[tag + ".rows"] = y16.cpu(), y32.cpu(), rows
    if K %% 128 == 0:        # the fp8 inference GEMMs: wscale + bias, the row factor (BF16) and the bf16 residual update (RES16)
        a8 = torch.randn((rows, K), device="cuda").to(ops.FP8)
        w8, sc = ops.quantize_fp8_rows(torch.randn((N, K), device="cuda") * 0.05)
        z16 = torch.empty((rows, N), device="cuda", dtype=torch.bfloat16)
        ops.gemm_fp8(a8, w8, sc, ops.EPI_BF16, z16, bias=bias, af=af, ntok=ntok)
        r16 = torch.empty((rows, N), device="cuda", dtype=torch.bfloat16)
        ops.gemm_fp8(a8, w8, sc, ops.EPI_RES16, r16, bias=bias, resid=resid.to(torch.bfloat16), af=af, vec=vec, bt=bt, ntok=ntok)
        out[tag + ".fp8_bf16"], out[tag + ".fp8_res16"] = z16.cpu(), r16.cpu()
torch.save(out, sys.argv[1])
''' % ROOT


def _run(peel: str, path: str):
    env = dict(os.environ, AIM_GEMM_PEEL=peel)
    subprocess.run([sys.executable, "-c", CHILD, path], check=True, env=env, timeout=300)
    return torch.load(path, weights_only=True)


def test_peeled_launch_is_bit_identical(tmp_path):
    a = _run("1", str(tmp_path / "peel.pt"))
    b = _run("0", str(tmp_path / "nopeel.pt"))
    assert set(a) == set(b)
    for k in a:
        if k.endswith(".rows"):
            assert a[k] == b[k] and a[k] % 256 != 0
        else:
            assert torch.isfinite(a[k].float()).all(), k
            assert torch.equal(a[k], b[k]), (k, (a[k].float() - b[k].float()).abs().max())
