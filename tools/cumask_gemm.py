"""Large-GEMM time on a CU-masked stream vs an ordinary one (hipExtStreamCreateWithCUMask).

Measured: keeping 8 CUs (one per XCD) out of the stream costs the persistent GEMM +41 % (0.426 -> 0.602 ms), not 3 %: workgroups
are striped over XCDs and shader engines in launch order and WAIT for a CU of their engine even when other engines have one free
(tools/cumask_probe.hip shows the mask -> CU map).  The same striping is why a side stream starves beside a persistent grid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops
import ctypes


def masked_stream(dev, words):
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr) == 0
    return torch.cuda.ExternalStream(st.value, device=dev)

dev = torch.device("cuda:0")
M = 100864
a = torch.randn((M, 768), device=dev).to(torch.bfloat16)
w = (torch.randn((2304, 768), device=dev) * 0.03).to(torch.bfloat16)
out = torch.empty((M, 2304), dtype=torch.bfloat16, device=dev)
w2 = (torch.randn((3264, 768), device=dev) * 0.03).to(torch.bfloat16)
out2 = torch.empty((M, 3264), dtype=torch.bfloat16, device=dev); out3 = torch.empty_like(out2)
def t(fn, st, n=10):
    with torch.cuda.stream(st):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n): fn()
        e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
plain = torch.cuda.Stream(device=dev)
masks = {"248 (all but cu0 of se0 per xcd)": [0xFFFFFF00] + [0xFFFFFFFF] * 7, "256 (full mask)": [0xFFFFFFFF] * 8,
         "224 (one per SE out?)": [0xFFFFFF00, 0xFFFFFFFF, 0xFFFFFF00, 0xFFFFFFFF, 0xFFFFFF00, 0xFFFFFFFF, 0xFFFFFF00, 0xFFFFFFFF]}
print("plain stream: qkv %.3f ms  fc1(act) %.3f ms" % (t(lambda: ops.gemm(a, w, ops.EPI_BF16, out), plain),
      t(lambda: ops.gemm(a, w2, ops.EPI_ACT, out2, out2=out3), plain)))
for name, m in masks.items():
    st = masked_stream(dev, m)
    print("mask %s: qkv %.3f ms  fc1(act) %.3f ms" % (name, t(lambda: ops.gemm(a, w, ops.EPI_BF16, out), st),
          t(lambda: ops.gemm(a, w2, ops.EPI_ACT, out2, out2=out3), st)))
