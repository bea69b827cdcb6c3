"""Thin tensor-level wrappers over the C ABI (include/aim_kernels.h).

PyTorch is used for device memory and streams only; every function here launches exactly the HIP
kernel(s) of the same name on the current stream and returns without synchronising.
"""
from ctypes import byref
from typing import Optional

import torch

from .lib import GemmArgs, check, load_library

EPI_BF16, EPI_ACT, EPI_DACT, EPI_F32, EPI_EXPSUM, EPI_ACT8, EPI_RES16 = 0, 1, 2, 3, 4, 5, 6
FP8 = torch.float8_e4m3fn          # OCP e4m3 (gfx950); stored as bytes
ACT_QGELU, ACT_GELU = 0, 1

BF16 = torch.bfloat16
F32 = torch.float32


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The raw hipStream_t of torch's current stream on the current device.  (`torch.cuda.current_stream()` builds a Stream
    object behind several Python-level device lookups: 8 us a call, 5 ms per forward at one sample x 3 views of ViT-L/14.)"""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(_RAW_DEVICE())
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional HIP-event timing of the large GEMM launches (bench.py roofline leg).  Events are
    recorded on the stream the kernel is launched on; durations are read after the timed region."""

    def __init__(self, min_flops: float = 1e10):
        self.min_flops = min_flops
        self.enabled = False
        self.records = []          # (epi, flops, start_event, end_event)

    def start(self):
        self.records, self.enabled = [], True

    def stop(self):
        self.enabled = False

    def summary(self):
        """-> {epi: dict(launches, flops, ms)} ; call after torch.cuda.synchronize()."""
        out = {}
        for epi, fl, e0, e1 in self.records:
            d = out.setdefault(epi, dict(launches=0, flops=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += fl
            d["ms"] += e0.elapsed_time(e1)
        return out


GEMM_TIMER = KernelTimer()


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: Optional[torch.Tensor], dtype, name: str):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")


def frag_elems(M: int, N: int) -> int:
    """Elements of the fragment-ordered side buffer of an ``[M, N]`` ACT / DACT pair (aim_gemm_args.aux_frag)."""
    return ((M + 255) // 256) * ((N + 255) // 256) * 65536


def frag_buffer(M: int, N: int, device) -> torch.Tensor:
    return torch.empty((frag_elems(M, N),), dtype=BF16, device=device)


def gemm(a: torch.Tensor, w: torch.Tensor, epi: int, out: torch.Tensor, *, bias=None, resid=None,
         af=None, at=None, vec=None, bt=None, ntok: int = 0, aux=None, out2=None, act: int = 0,
         scale: float = 1.0, rs_bias_only: bool = False, batch: int = 1, stride_a: int = 0,
         stride_w: int = 0, M: Optional[int] = None, N: Optional[int] = None, K: Optional[int] = None,
         lda: Optional[int] = None, ldw: Optional[int] = None, ldv: Optional[int] = None, n_split: int = 0,
         act2: int = 0, xrow=None, reserve_cus: int = 0, probe=None, aux_grad: bool = False, aux_frag: bool = False,
         slot_stride: int = 0):
    """``out = epilogue(a @ w.T)``; ``a`` is ``[M, K]`` (row stride ``lda``), ``w`` is ``[N, K]``."""
    lib = load_library()
    _chk(a, BF16, "a"); _chk(w, BF16, "w")
    for n_, t_ in (("bias", bias), ("resid", resid), ("af", af), ("at", at), ("vec", vec), ("bt", bt)):
        _chk(t_, F32, n_)
    _chk(aux, BF16, "aux"); _chk(out2, BF16, "out2")
    g = GemmArgs()
    g.A, g.W = a.data_ptr(), w.data_ptr()
    g.M = a.shape[0] if M is None else M
    g.K = a.shape[1] if K is None else K
    g.N = w.shape[0] if N is None else N
    g.lda = a.stride(0) if lda is None else lda
    g.ldw = w.stride(0) if ldw is None else ldw
    g.strideA, g.strideW = stride_a, stride_w
    g.bias, g.resid = _p(bias), _p(resid)
    g.ldr = resid.stride(0) if resid is not None else 0
    g.af, g.at, g.vec, g.bt = _p(af), _p(at), _p(vec), _p(bt)
    g.ldv = (vec.stride(0) if ldv is None else ldv) if vec is not None else 0
    g.ntok = ntok
    g.aux = _p(aux)
    g.ldaux = aux.stride(0) if aux is not None else 0
    g.out = out.data_ptr()
    # EXPSUM: ldo = float stride between the (max, sum) slot groups of consecutive tiles (0: the tile's own 16 | 32)
    g.ldo = (out.stride(0) if out.dim() >= 2 else 0) if epi != EPI_EXPSUM else int(slot_stride)
    g.out2 = _p(out2)
    g.ldo2 = out2.stride(0) if out2 is not None else 0
    g.scale, g.act, g.rs_bias_only = scale, act, int(rs_bias_only)
    g.n_split, g.act2 = n_split, act2
    _chk(xrow, BF16, "xrow")
    g.xrow = _p(xrow)
    g.ldx = xrow.stride(0) if xrow is not None else 0
    g.reserve_cus = int(reserve_cus)       # per-call: CUs this persistent launch leaves to other streams
    g.aux_grad = int(bool(aux_grad))       # ACT: out2 = act'(pre) instead of pre; DACT: aux holds act'(pre)
    g.aux_frag = int(bool(aux_frag))       # ACT / DACT: out2 / aux is a fragment-ordered buffer (frag_buffer)
    if aux_frag:
        t_ = out2 if epi == EPI_ACT else aux
        if epi not in (EPI_ACT, EPI_DACT) or batch != 1 or g.M < 1024 or t_ is None or not t_.is_contiguous() \
                or t_.numel() < frag_elems(g.M, g.N):
            raise ValueError("aux_frag: ACT / DACT on the large-tile kernel (M >= 1024, batch 1) with a frag_buffer(M, N)")
        g.ldo2 = g.ldaux = 0               # (no row stride: the buffer is fragment-ordered)
    if probe is not None:                  # diagnostics (tools/probe_gemm.py): [cap, 4] int64 device tensor
        g.probe, g.probe_cap = probe.data_ptr(), probe.shape[0]
    if epi in (EPI_BF16, EPI_ACT, EPI_DACT):
        _chk(out, BF16, "out")
    else:
        _chk(out, F32, "out")
    timed = GEMM_TIMER.enabled and 2.0 * g.M * g.N * g.K * batch >= GEMM_TIMER.min_flops
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.aim_gemm_bf16(byref(g), epi, batch, _stream()), "aim_gemm_bf16")
    if timed:
        e1.record()
        GEMM_TIMER.records.append((epi, 2.0 * g.M * g.N * g.K * batch, e0, e1))
    return out


def gemm_fp8(a8: torch.Tensor, w8: torch.Tensor, wscale: torch.Tensor, epi: int, out: torch.Tensor, *, bias=None,
             resid=None, af=None, at=None, vec=None, bt=None, ntok: int = 0, act: int = 0, n_split: int = 0, act2: int = 0,
             ldv: Optional[int] = None, reserve_cus: int = 0):
    """``out = epilogue(wscale[n] * (a8 @ w8.T))``: fp8 e4m3 operands on the block-scaled MFMA (inference only).
    ``a8`` [M, K], ``w8`` [N, K] are float8_e4m3fn (or uint8 views); ``out`` bf16 (EPI_BF16), f32 (EPI_F32) or fp8
    (EPI_ACT8)."""
    lib = load_library()
    for n_, t_ in (("a8", a8), ("w8", w8)):
        if not t_.is_cuda or t_.element_size() != 1 or t_.stride(-1) != 1:
            raise TypeError(f"{n_}: expected a GPU fp8 (1-byte) tensor with a contiguous last dimension")
    for n_, t_ in (("wscale", wscale), ("bias", bias), ("af", af), ("at", at), ("vec", vec), ("bt", bt)):
        _chk(t_, F32, n_)
    _chk(resid, BF16 if epi == EPI_RES16 else F32, "resid")       # RES16: the bf16 residual stream of the fp8 inference path
    g = GemmArgs()
    g.A, g.W = a8.data_ptr(), w8.data_ptr()
    g.M, g.K, g.N = a8.shape[0], a8.shape[1], w8.shape[0]
    g.lda, g.ldw = a8.stride(0), w8.stride(0)
    g.bias, g.resid, g.wscale = _p(bias), _p(resid), wscale.data_ptr()
    g.ldr = resid.stride(0) if resid is not None else 0
    g.af, g.at, g.vec, g.bt = _p(af), _p(at), _p(vec), _p(bt)
    g.ldv = (vec.stride(0) if ldv is None else ldv) if vec is not None else 0
    g.ntok = ntok
    g.out, g.ldo = out.data_ptr(), out.stride(0)
    g.act, g.n_split, g.act2, g.scale = act, n_split, act2, 1.0
    g.reserve_cus = int(reserve_cus)
    if epi in (EPI_BF16, EPI_RES16):
        _chk(out, BF16, "out")
    elif epi == EPI_F32:
        _chk(out, F32, "out")
    elif out.element_size() != 1:
        raise TypeError("out: EPI_ACT8 writes fp8 bytes")
    timed = GEMM_TIMER.enabled and 2.0 * g.M * g.N * g.K >= GEMM_TIMER.min_flops
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.aim_gemm_fp8(byref(g), epi, _stream()), "aim_gemm_fp8")
    if timed:
        e1.record()
        GEMM_TIMER.records.append((100 + epi, 2.0 * g.M * g.N * g.K, e0, e1))
    return out


def quantize_fp8_rows(w: torch.Tensor):
    """Per-output-channel fp8 quantisation of a weight ``[N, K]`` (one-off host-side staging, not a hot-path op):
    ``scale[n] = amax_n / 448``, ``w8 = e4m3(w / scale)``.  Returns (w8 as float8_e4m3fn on w's device, scale f32)."""
    wf = w.detach().float()
    amax = wf.abs().amax(dim=1).clamp_min(1e-12)
    scale = amax / 448.0
    q = (wf / scale[:, None]).clamp_(-448.0, 448.0)
    # the cast itself runs on the CPU: every torch build rounds f32 -> e4m3fn to nearest-even there
    w8 = q.cpu().to(FP8).to(w.device)
    return w8.contiguous(), scale.contiguous()


def layernorm_fwd_fp8(x, gamma, beta, rows, D, ldx, y8, ldy=None, eps: float = 1e-5):
    _chk(x, F32, "x"); _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta")
    if y8.element_size() != 1 or not y8.is_cuda:
        raise TypeError("y8: expected a GPU fp8 (1-byte) tensor")
    check(load_library().aim_layernorm_fwd_fp8(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), y8.data_ptr(),
                                               D if ldy is None else ldy, rows, D, eps, _stream()), "aim_layernorm_fwd_fp8")


def layernorm_fwd_x16(x, gamma, beta, rows, D, ldx, *, y_bf16=None, y_f32=None, y8=None, ldy=None, eps: float = 1e-5):
    """LayerNorm over bf16 rows (the fp8 inference path's residual stream); any subset of bf16 / f32 / fp8 outputs."""
    _chk(x, BF16, "x"); _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta"); _chk(y_bf16, BF16, "y_bf16"); _chk(y_f32, F32, "y_f32")
    if y8 is not None and (y8.element_size() != 1 or not y8.is_cuda):
        raise TypeError("y8: expected a GPU fp8 (1-byte) tensor")
    check(load_library().aim_layernorm_fwd_x16(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), _p(y_bf16), _p(y_f32), _p(y8),
                                               D if ldy is None else ldy, rows, D, eps, _stream()), "aim_layernorm_fwd_x16")


def attn_fwd_fp8(qkv, out8, BT, N, H, lse=None):
    _chk(qkv, BF16, "qkv"); _chk(lse, F32, "lse")
    if out8.element_size() != 1 or not out8.is_cuda:
        raise TypeError("out8: expected a GPU fp8 (1-byte) tensor")
    check(load_library().aim_attn_fwd_fp8(qkv.data_ptr(), out8.data_ptr(), _p(lse), BT, N, H, _stream()), "aim_attn_fwd_fp8")


def expsum_tiles(M: int, N: int) -> int:
    return load_library().aim_gemm_expsum_tiles(M, N)


def wgrad(g: torch.Tensor, a: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor] = None, at=None, ntok: int = 0):
    """``dw[n,k] += sum_m g[m,n] a[m,k]``; ``db[n] += sum_m at[m % ntok] g[m,n]`` (``at`` None: 1) in the same pass
    (fp32 accumulate in place; split-M partial slabs summed in a fixed order: no atomics)."""
    _chk(g, BF16, "g"); _chk(a, BF16, "a"); _chk(dw, F32, "dw"); _chk(db, F32, "db"); _chk(at, F32, "at")
    assert g.shape[0] == a.shape[0] and dw.shape == (g.shape[1], a.shape[1])
    lib = load_library()
    nbytes = lib.aim_wgrad_workspace_bytes(g.shape[0], g.shape[1], a.shape[1])
    ws = torch.empty(nbytes // 4, dtype=F32, device=g.device) if nbytes else None
    check(lib.aim_wgrad_bias_bf16(g.data_ptr(), g.stride(0), a.data_ptr(), a.stride(0), dw.data_ptr(), dw.stride(0), _p(db),
                                  _p(at), ntok, g.shape[0], g.shape[1], a.shape[1], _p(ws), nbytes, _stream()), "aim_wgrad_bias_bf16")


def layernorm_fwd(x, gamma, beta, rows, D, ldx, *, y_bf16=None, y_f32=None, ldy=None, mean=None, rstd=None,
                  eps: float = 1e-5):
    _chk(x, F32, "x"); _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta")
    _chk(y_bf16, BF16, "y_bf16"); _chk(y_f32, F32, "y_f32"); _chk(mean, F32, "mean"); _chk(rstd, F32, "rstd")
    check(load_library().aim_layernorm_fwd(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), _p(y_bf16),
                                           _p(y_f32), D if ldy is None else ldy, _p(mean), _p(rstd), rows, D,
                                           eps, _stream()), "aim_layernorm_fwd")


def layernorm_bwd(dy, x, gamma, mean, rstd, rows, D, *, lddy, ldx, lddx, dres=None, dx=None, dx_bf16=None,
                  dgamma=None, dbeta=None):
    for n_, t_ in (("x", x), ("gamma", gamma), ("mean", mean), ("rstd", rstd),
                   ("dx", dx), ("dgamma", dgamma), ("dbeta", dbeta)):
        _chk(t_, F32, n_)
    _chk(dx_bf16, BF16, "dx_bf16")
    _chk(dy, dy.dtype if dy.dtype in (F32, BF16) else F32, "dy")
    if dres is not None:
        _chk(dres, dres.dtype if dres.dtype in (F32, BF16) else F32, "dres")
    check(load_library().aim_layernorm_bwd(dy.data_ptr(), int(dy.dtype == BF16), lddy, x.data_ptr(), ldx, gamma.data_ptr(),
                                           mean.data_ptr(), rstd.data_ptr(), _p(dres),
                                           int(dres is not None and dres.dtype == BF16), _p(dx), _p(dx_bf16), lddx,
                                           _p(dgamma), _p(dbeta), rows, D, _stream()), "aim_layernorm_bwd")


LN_FSUM_GROUPS = 13     # token groups per frame of layernorm_bwd_fsum's partial sums (16 tokens at 197: 4 rows per wave;
                        # with 4 groups the 2 048 workgroups ran as one full round and a nearly empty one)


def layernorm_bwd_fsum(dy, x, gamma, mean, rstd, dres, dx_bf16, w, partial, frames, ntok, D):
    """layernorm_bwd for bf16 dy / dres / dx (contiguous rows) that also writes partial[frames, LN_FSUM_GROUPS, D]: the
    per-frame sums of w[n] * dx over each token group; ``frame_sum(partial, None, out, frames, LN_FSUM_GROUPS, D)`` finishes."""
    _chk(dy, BF16, "dy"); _chk(dres, BF16, "dres"); _chk(dx_bf16, BF16, "dx_bf16")
    for n_, t_ in (("x", x), ("gamma", gamma), ("mean", mean), ("rstd", rstd), ("w", w), ("partial", partial)):
        _chk(t_, F32, n_)
    assert partial.numel() >= frames * LN_FSUM_GROUPS * D
    check(load_library().aim_layernorm_bwd_fsum(dy.data_ptr(), D, x.data_ptr(), D, gamma.data_ptr(), mean.data_ptr(),
                                                rstd.data_ptr(), dres.data_ptr(), dx_bf16.data_ptr(), D, _p(w),
                                                partial.data_ptr(), LN_FSUM_GROUPS, frames, ntok, D, _stream()),
          "aim_layernorm_bwd_fsum")


def attn_fwd(qkv, out, lse, BT, N, H):
    _chk(qkv, BF16, "qkv"); _chk(out, BF16, "out"); _chk(lse, F32, "lse")
    check(load_library().aim_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), BT, N, H, _stream()),
          "aim_attn_fwd")


def attn_bwd(qkv, out, dout, lse, delta, dqkv, BT, N, H):
    _chk(qkv, BF16, "qkv"); _chk(out, BF16, "out"); _chk(dout, BF16, "dout"); _chk(dqkv, BF16, "dqkv")
    _chk(lse, F32, "lse"); _chk(delta, F32, "delta")
    check(load_library().aim_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                      delta.data_ptr(), dqkv.data_ptr(), BT, N, H, _stream()), "aim_attn_bwd")


def cls_attn_fwd(qkv, out_cls, probs, B, T, N, H):
    _chk(qkv, BF16, "qkv"); _chk(out_cls, BF16, "out_cls"); _chk(probs, F32, "probs")
    check(load_library().aim_cls_attn_fwd(qkv.data_ptr(), out_cls.data_ptr(), probs.data_ptr(), B, T, N, H,
                                          _stream()), "aim_cls_attn_fwd")


def cls_attn_bwd(qkv, probs, dout_cls, dqkv, B, T, N, H, compact: bool = False):
    """compact=False: add into the class rows of dqkv [B*T*N, 3D]; compact=True: write dqkv [B*T, 3D]."""
    _chk(qkv, BF16, "qkv"); _chk(dout_cls, BF16, "dout_cls"); _chk(dqkv, BF16, "dqkv"); _chk(probs, F32, "probs")
    check(load_library().aim_cls_attn_bwd(qkv.data_ptr(), probs.data_ptr(), dout_cls.data_ptr(), dqkv.data_ptr(),
                                          int(compact), B, T, N, H, _stream()), "aim_cls_attn_bwd")


def tattn_fwd(qkv, out, probs, B, T, N, H):
    """Temporal attention over the T frames of every token position (stock-AIM block): out [M, D], probs [B*N, H, T, T]."""
    _chk(qkv, BF16, "qkv"); _chk(out, BF16, "out"); _chk(probs, F32, "probs")
    check(load_library().aim_tattn_fwd(qkv.data_ptr(), out.data_ptr(), probs.data_ptr(), B, T, N, H, _stream()), "aim_tattn_fwd")


def tattn_bwd(qkv, probs, dout, dqkv, B, T, N, H):
    _chk(qkv, BF16, "qkv"); _chk(probs, F32, "probs"); _chk(dout, BF16, "dout"); _chk(dqkv, BF16, "dqkv")
    check(load_library().aim_tattn_bwd(qkv.data_ptr(), probs.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), B, T, N, H, _stream()),
          "aim_tattn_bwd")


def add_bf16(a, b, out):
    """out = a + b (bf16, 2-D, row-strided)."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(out, BF16, "out")
    R, C = a.shape
    check(load_library().aim_add_bf16(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), R, C,
                                      _stream()), "aim_add_bf16")
    return out


def acc_bf16(x, s):
    """x (f32, dense 2-D) += s (bf16, row-strided)."""
    _chk(x, F32, "x"); _chk(s, BF16, "s")
    assert x.is_contiguous() and x.shape == s.shape
    check(load_library().aim_acc_bf16(x.data_ptr(), s.data_ptr(), s.stride(0), x.shape[0], x.shape[1], _stream()), "aim_acc_bf16")
    return x


def lambda_partials(partials, lam, one_minus, BT):
    """lamda from [BT, 16, 2] (max, sum) slots: 8 ``ow`` partials, 8 ``cw`` partials (EXPSUM GEMM with ``xrow``)."""
    _chk(partials, F32, "partials"); _chk(lam, F32, "lam"); _chk(one_minus, F32, "one_minus")
    check(load_library().aim_lambda_partials(partials.data_ptr(), lam.data_ptr(), _p(one_minus), BT, _stream()),
          "aim_lambda_partials")


def qk_cross(qkv, kx, ss, BT, N, D, scale):
    """ss[bt, i] = scale * q_i . kx[bt] (full width D): the pass over q of lamda's cw statistic."""
    _chk(qkv, BF16, "qkv"); _chk(kx, BF16, "kx"); _chk(ss, F32, "ss")
    check(load_library().aim_qk_cross(qkv.data_ptr(), kx.data_ptr(), kx.stride(0), ss.data_ptr(), BT, N, D, scale,
                                      _stream()), "aim_qk_cross")


def qk_border(qkv, kx, ss, partials, slot0: int, BT, N, D, scale):
    """Border of the lamda statistics for N = 257 (see aim_qk_border): fills ss [BT, N] and partials[:, slot0:slot0 + 2]."""
    _chk(qkv, BF16, "qkv"); _chk(kx, BF16, "kx"); _chk(ss, F32, "ss"); _chk(partials, F32, "partials")
    assert partials.dim() == 3 and partials.shape[0] == BT and partials.shape[2] == 2 and partials.is_contiguous()
    check(load_library().aim_qk_border(qkv.data_ptr(), kx.data_ptr(), kx.stride(0), ss.data_ptr(), partials.data_ptr(), slot0,
                                       partials.shape[1], BT, N, D, scale, _stream()), "aim_qk_border")


def lambda_(qkv, kx, partials, ntiles, lam, one_minus, BT, N, D, scale, ss=None):
    """lamda = cw / (cw + ow); ``ss`` = precomputed cross scores (qk_cross), else computed here from q and kx."""
    _chk(qkv, BF16, "qkv"); _chk(kx, BF16, "kx"); _chk(partials, F32, "partials"); _chk(lam, F32, "lam")
    _chk(one_minus, F32, "one_minus"); _chk(ss, F32, "ss")
    check(load_library().aim_lambda(qkv.data_ptr(), kx.data_ptr(), kx.stride(0), _p(ss), partials.data_ptr(), ntiles,
                                    lam.data_ptr(), _p(one_minus), BT, N, D, scale, _stream()), "aim_lambda")


_IN_DTYPES = {torch.float32: 0, torch.uint8: 1, torch.bfloat16: 2}


def patchify(imgs, A, B, T, H, W, p, Kp, mean3=None, std3=None):
    if imgs.dtype not in _IN_DTYPES:
        raise TypeError(f"patchify: unsupported input dtype {imgs.dtype} (float32, uint8, bfloat16)")
    if not imgs.is_cuda or not imgs.is_contiguous():
        raise ValueError("patchify: imgs must be a contiguous GPU tensor")
    _chk(A, BF16, "A"); _chk(mean3, F32, "mean3"); _chk(std3, F32, "std3")
    check(load_library().aim_patchify(imgs.data_ptr(), _IN_DTYPES[imgs.dtype], _p(mean3), _p(std3), A.data_ptr(),
                                      B, T, H, W, p, Kp, _stream()), "aim_patchify")


def embed_ln(tok, cls, pos, temporal, gamma, beta, x, mean, rstd, B, T, N, D, eps=1e-5):
    _chk(tok, BF16, "tok")
    for n_, t_ in (("cls", cls), ("pos", pos), ("temporal", temporal), ("gamma", gamma), ("beta", beta), ("x", x),
                   ("mean", mean), ("rstd", rstd)):
        _chk(t_, F32, n_)
    check(load_library().aim_embed_ln(tok.data_ptr(), cls.data_ptr(), pos.data_ptr(), temporal.data_ptr(),
                                      gamma.data_ptr(), beta.data_ptr(), x.data_ptr(), mean.data_ptr(),
                                      rstd.data_ptr(), B, T, N, D, eps, _stream()), "aim_embed_ln")


def embed_bwd(dx, tok, cls, pos, temporal, gamma, mean, rstd, dtemporal, B, T, N, D):
    _chk(tok, BF16, "tok")
    for n_, t_ in (("cls", cls), ("pos", pos), ("temporal", temporal), ("gamma", gamma), ("mean", mean),
                   ("rstd", rstd), ("dtemporal", dtemporal)):
        _chk(t_, F32, n_)
    _chk(dx, dx.dtype if dx.dtype in (F32, BF16) else F32, "dx")
    lib = load_library()
    nbytes = lib.aim_embed_bwd_workspace_bytes(B, T, N, D)
    ws = torch.empty((nbytes // 4,), dtype=F32, device=dx.device)      # two-stage, bitwise reproducible reduction
    check(lib.aim_embed_bwd(dx.data_ptr(), int(dx.dtype == BF16), tok.data_ptr(), cls.data_ptr(), pos.data_ptr(),
                            temporal.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                            dtemporal.data_ptr(), B, T, N, D, ws.data_ptr(), nbytes, _stream()), "aim_embed_bwd")


def frame_sum(x, w, out, frames, ntok, D):
    _chk(x, x.dtype if x.dtype in (F32, BF16) else F32, "x"); _chk(w, F32, "w"); _chk(out, F32, "out")
    check(load_library().aim_frame_sum(x.data_ptr(), int(x.dtype == BF16), _p(w), out.data_ptr(), frames, ntok, D, _stream()),
          "aim_frame_sum")


def colsum(X, out, *, af=None, at=None, ntok=0):
    _chk(X, BF16, "X"); _chk(out, F32, "out"); _chk(af, F32, "af"); _chk(at, F32, "at")
    ws = None
    if X.shape[0] > 64:             # two-stage through a scratch buffer: no atomics, bitwise reproducible
        ws = torch.empty((min(1024, (X.shape[0] + 63) // 64), X.shape[1]), dtype=F32, device=X.device)
    check(load_library().aim_colsum_bf16(X.data_ptr(), X.stride(0), _p(af), _p(at), ntok, out.data_ptr(),
                                         X.shape[0], X.shape[1], _p(ws), ws.numel() * 4 if ws is not None else 0,
                                         _stream()), "aim_colsum_bf16")


def cast_bf16(src, dst, transpose=False):
    """``dst = bf16(src)`` or ``bf16(src.T)``; ``dst`` may be a column slice of a wider matrix (row-strided)."""
    _chk(src, F32, "src"); _chk(dst, BF16, "dst")
    R, C = src.shape
    assert tuple(dst.shape) == ((C, R) if transpose else (R, C)) and src.is_contiguous()
    check(load_library().aim_cast_bf16(src.data_ptr(), dst.data_ptr(), R, C, int(transpose), dst.stride(0), _stream()),
          "aim_cast_bf16")


def scale_rows(x, s, y=None, y_f32=None):
    _chk(x, F32, "x"); _chk(s, F32, "s"); _chk(y, BF16, "y"); _chk(y_f32, F32, "y_f32")
    R, C = x.shape
    check(load_library().aim_scale_rows(x.data_ptr(), s.data_ptr(), _p(y), _p(y_f32), R, C, _stream()),
          "aim_scale_rows")


def add_rows(dst, dst_row_stride: int, src):
    """dst[r * dst_row_stride + c] += src[r, c]  (bf16 += fp32)."""
    _chk(dst, BF16, "dst"); _chk(src, F32, "src")
    R, C = src.shape
    check(load_library().aim_add_rows_bf16(dst.data_ptr(), int(dst_row_stride), src.data_ptr(), R, C, _stream()),
          "aim_add_rows_bf16")


def adamw_flat(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale: float = 1.0):
    for n_, t_ in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t_, F32, n_)
    check(load_library().aim_adamw_flat(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1,
                                        beta2, eps, weight_decay, step, grad_scale, _stream()), "aim_adamw_flat")


def head_fwd(feat, drop, W, bias):
    """feat [B, T, D] f32 -> (pooled [B, D], score [B, C]) ; drop [B, D] factor table or None."""
    _chk(feat, F32, "feat"); _chk(drop, F32, "drop"); _chk(W, F32, "W"); _chk(bias, F32, "bias")
    B, T, D = feat.shape
    C = W.shape[0]
    assert feat.is_contiguous() and W.is_contiguous() and W.shape[1] == D
    pooled = torch.empty((B, D), dtype=F32, device=feat.device)
    score = torch.empty((B, C), dtype=F32, device=feat.device)
    check(load_library().aim_head_fwd(feat.data_ptr(), _p(drop), W.data_ptr(), _p(bias), pooled.data_ptr(),
                                      score.data_ptr(), B, T, D, C, _stream()), "aim_head_fwd")
    return pooled, score


def head_bwd(dscore, pooled, drop, W, T, need_dfeat=True):
    """-> (dW [C, D], db [C], dfeat [B, T, D] or None)."""
    _chk(dscore, F32, "dscore"); _chk(pooled, F32, "pooled"); _chk(drop, F32, "drop"); _chk(W, F32, "W")
    B, C = dscore.shape
    D = W.shape[1]
    assert dscore.is_contiguous() and pooled.is_contiguous()
    dW = torch.zeros((C, D), dtype=F32, device=W.device)
    db = torch.zeros((C,), dtype=F32, device=W.device)
    dfeat = torch.empty((B, T, D), dtype=F32, device=W.device) if need_dfeat else None
    check(load_library().aim_head_bwd(dscore.data_ptr(), pooled.data_ptr(), _p(drop), W.data_ptr(), dW.data_ptr(),
                                      db.data_ptr(), _p(dfeat), B, T, D, C, _stream()), "aim_head_bwd")
    return dW, db, dfeat


def ce_topk(score, label, k2: int = 5, need_grad: bool = True):
    """-> (out3 = [mean CE, top-1, top-k2] f32, dscore [B, C] = (softmax - onehot) / B or None)."""
    _chk(score, F32, "score")
    if label.dtype != torch.int64 or not label.is_cuda:
        raise TypeError("ce_topk: label must be an int64 GPU tensor")
    B, C = score.shape
    assert score.is_contiguous() and label.numel() == B
    dscore = torch.empty_like(score) if need_grad else None
    ws = torch.empty((B, 4), dtype=F32, device=score.device)
    out3 = torch.empty((3,), dtype=F32, device=score.device)
    check(load_library().aim_ce_topk(score.data_ptr(), label.data_ptr(), _p(dscore), ws.data_ptr(), out3.data_ptr(), B, C,
                                     k2, _stream()), "aim_ce_topk")
    return out3, dscore


class CastTable:
    """Device-resident table of (fp32 src -> bf16 dst [transposed]) casts, run in one launch."""

    def __init__(self, entries, device):
        import struct
        self.keep = entries            # keep the tensors alive: the table holds raw pointers
        raw = b"".join(struct.pack("<QQiiii", src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1], dst.stride(0),
                                   int(tr)) for src, dst, tr in entries)
        for src, dst, tr in entries:       # tr: False / True = bf16 cast (plain / transposed); 2 = fp32 copy
            _chk(src, F32, "src"); _chk(dst, F32 if tr == 2 else BF16, "dst")
            assert src.dim() == 2 and src.is_contiguous() and src.numel() < 2 ** 31
            assert tuple(dst.shape) == ((src.shape[1], src.shape[0]) if tr is True or tr == 1 else tuple(src.shape))
        self.n = len(entries)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.ptrs = tuple(src.data_ptr() for src, _, _ in entries)

    def run(self):
        check(load_library().aim_cast_multi(self.table.data_ptr(), self.n, _stream()), "aim_cast_multi")


# ---- reference-precision (fp32) forward path: include/aim_kernels.h, "Reference-precision" section ----------------------
def gemm_f32(a: torch.Tensor, w: torch.Tensor, epi: int, out: torch.Tensor, *, bias=None, resid=None, af=None, at=None,
             vec=None, bt=None, ntok: int = 0, act: int = 0, n_split: int = 0, act2: int = 0, ldv: Optional[int] = None,
             batch: int = 1, stride_a: int = 0, stride_w: int = 0, M: Optional[int] = None, N: Optional[int] = None,
             K: Optional[int] = None, lda: Optional[int] = None, ldw: Optional[int] = None, ldo: Optional[int] = None,
             out2=None, aux=None):
    """``out(f32) = epilogue(a @ w.T)`` with fp32 operands on the f32 MFMA; ``epi`` in (EPI_BF16 = linear, EPI_ACT, EPI_DACT,
    EPI_F32).  ``out2`` (EPI_ACT): receives the f32 pre-activation; ``aux`` (EPI_DACT): that pre-activation."""
    for n_, t_ in (("a", a), ("w", w), ("out", out), ("bias", bias), ("resid", resid), ("af", af), ("at", at), ("vec", vec), ("bt", bt),
                   ("out2", out2), ("aux", aux)):
        _chk(t_, F32, n_)
    g = GemmArgs()
    g.A, g.W = a.data_ptr(), w.data_ptr()
    g.M = a.shape[0] if M is None else M
    g.K = a.shape[1] if K is None else K
    g.N = w.shape[0] if N is None else N
    g.lda = a.stride(0) if lda is None else lda
    g.ldw = w.stride(0) if ldw is None else ldw
    g.strideA, g.strideW = stride_a, stride_w
    g.bias, g.resid = _p(bias), _p(resid)
    g.ldr = resid.stride(0) if resid is not None else 0
    g.af, g.at, g.vec, g.bt = _p(af), _p(at), _p(vec), _p(bt)
    g.ldv = (vec.stride(0) if ldv is None else ldv) if vec is not None else 0
    g.ntok = ntok
    g.out = out.data_ptr()
    g.ldo = (out.stride(-2) if out.dim() >= 2 else g.N) if ldo is None else ldo
    g.act, g.n_split, g.act2, g.scale = act, n_split, act2, 1.0
    if out2 is not None:
        g.out2, g.ldo2 = out2.data_ptr(), out2.stride(0)
    if aux is not None:
        g.aux, g.ldaux = aux.data_ptr(), aux.stride(0)
    check(load_library().aim_gemm_f32(byref(g), epi, batch, _stream()), "aim_gemm_f32")
    return out


def attn_fwd_f32(qkv, out, BT, N, H):
    _chk(qkv, F32, "qkv"); _chk(out, F32, "out")
    check(load_library().aim_attn_fwd_f32(qkv.data_ptr(), out.data_ptr(), BT, N, H, _stream()), "aim_attn_fwd_f32")


def cls_attn_fwd_f32(qkv, row_stride: int, out_cls, B, T, H):
    _chk(qkv, F32, "qkv"); _chk(out_cls, F32, "out_cls")
    check(load_library().aim_cls_attn_fwd_f32(qkv.data_ptr(), int(row_stride), out_cls.data_ptr(), B, T, H, _stream()),
          "aim_cls_attn_fwd_f32")


def lambda_f32(scores, qkv, kx, lam, one_minus, BT, N, D, scale):
    for n_, t_ in (("scores", scores), ("qkv", qkv), ("kx", kx), ("lam", lam), ("one_minus", one_minus)):
        _chk(t_, F32, n_)
    check(load_library().aim_lambda_f32(scores.data_ptr(), scores.stride(-2), qkv.data_ptr(), kx.data_ptr(), kx.stride(0),
                                        lam.data_ptr(), _p(one_minus), BT, N, D, scale, _stream()), "aim_lambda_f32")


def patchify_f32(imgs, A, B, T, H, W, p, Kp, mean3=None, std3=None):
    if imgs.dtype not in (torch.float32, torch.uint8):
        raise TypeError(f"patchify_f32: unsupported input dtype {imgs.dtype} (float32, uint8)")
    if not imgs.is_cuda or not imgs.is_contiguous():
        raise ValueError("patchify_f32: imgs must be a contiguous GPU tensor")
    _chk(A, F32, "A"); _chk(mean3, F32, "mean3"); _chk(std3, F32, "std3")
    check(load_library().aim_patchify_f32(imgs.data_ptr(), _IN_DTYPES[imgs.dtype], _p(mean3), _p(std3), A.data_ptr(), B, T, H, W,
                                          p, Kp, _stream()), "aim_patchify_f32")


def embed_ln_f32(tok, cls, pos, temporal, gamma, beta, x, B, T, N, D, eps=1e-5, pre=None, mean=None, rstd=None):
    """``pre`` / ``mean`` / ``rstd`` (all or none): ln_pre's input rows and statistics, for the backward."""
    for n_, t_ in (("tok", tok), ("cls", cls), ("pos", pos), ("temporal", temporal), ("gamma", gamma), ("beta", beta), ("x", x),
                   ("pre", pre), ("mean", mean), ("rstd", rstd)):
        _chk(t_, F32, n_)
    check(load_library().aim_embed_ln_f32(tok.data_ptr(), cls.data_ptr(), pos.data_ptr(), temporal.data_ptr(), gamma.data_ptr(),
                                          beta.data_ptr(), x.data_ptr(), _p(pre), _p(mean), _p(rstd), B, T, N, D, eps, _stream()),
          "aim_embed_ln_f32")


def attn_bwd_f32(qkv, dout, dqkv, BT, N, H):
    """dqkv [BT*N, 3D] (written) from the fused f32 qkv rows and d(out) [BT*N, D]; probabilities recomputed."""
    for n_, t_ in (("qkv", qkv), ("dout", dout), ("dqkv", dqkv)):
        _chk(t_, F32, n_)
    lib = load_library()
    nbytes = lib.aim_attn_bwd_f32_workspace_bytes(BT, N, H)
    ws = torch.empty((nbytes // 4,), dtype=F32, device=qkv.device)
    check(lib.aim_attn_bwd_f32(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), BT, N, H, ws.data_ptr(), nbytes, _stream()),
          "aim_attn_bwd_f32")


def cls_attn_bwd_f32(qkv, row_stride: int, dout_cls, dqkv, B, T, H):
    """d(out_cls) [B*T, D] -> ADDED into the class rows of dqkv (call after ``attn_bwd_f32``)."""
    for n_, t_ in (("qkv", qkv), ("dout_cls", dout_cls), ("dqkv", dqkv)):
        _chk(t_, F32, n_)
    check(load_library().aim_cls_attn_bwd_f32(qkv.data_ptr(), int(row_stride), dout_cls.data_ptr(), dqkv.data_ptr(), B, T, H,
                                              _stream()), "aim_cls_attn_bwd_f32")


def tattn_fwd_f32(qkv, out, B, T, N, H):
    """Stock-AIM temporal attention over the T frames of every token (fp32, frame-major fused qkv rows)."""
    _chk(qkv, F32, "qkv"); _chk(out, F32, "out")
    check(load_library().aim_tattn_fwd_f32(qkv.data_ptr(), out.data_ptr(), B, T, N, H, _stream()), "aim_tattn_fwd_f32")


def tattn_bwd_f32(qkv, dout, dqkv, B, T, N, H):
    """... its backward, ADDED into ``dqkv`` [B*T*N, 3D]."""
    for n_, t_ in (("qkv", qkv), ("dout", dout), ("dqkv", dqkv)):
        _chk(t_, F32, n_)
    check(load_library().aim_tattn_bwd_f32(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), B, T, N, H, _stream()),
          "aim_tattn_bwd_f32")


def wgrad_f32(g, a, dw, db=None, at=None, ntok: int = 0):
    """``dw [Nw, Kw] += g.T @ a``; ``db [Nw] += sum_m at[m % ntok] * g[m]`` (fp32; row-strided views allowed)."""
    for n_, t_ in (("g", g), ("a", a), ("dw", dw), ("db", db), ("at", at)):
        _chk(t_, F32, n_)
    M, Nw = g.shape
    Kw = a.shape[1]
    assert a.shape[0] == M and tuple(dw.shape) == (Nw, Kw) and dw.is_contiguous() and g.stride(1) == 1 and a.stride(1) == 1
    lib = load_library()
    nbytes = lib.aim_wgrad_f32_workspace_bytes(M, Nw, Kw)
    ws = torch.empty((nbytes // 4,), dtype=F32, device=g.device)
    check(lib.aim_wgrad_f32(g.data_ptr(), g.stride(0), a.data_ptr(), a.stride(0), dw.data_ptr(), M, Nw, Kw, _p(db), _p(at), ntok,
                            ws.data_ptr(), nbytes, _stream()), "aim_wgrad_f32")
