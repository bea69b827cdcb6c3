"""``ViT_CLIP`` backbone (AIM ViT-CLIP + Adapters) on the HIP kernels of libaim_hip.so.

Drop-in for ``mmaction/models/backbones/vit_clip.py:327-458`` of the reference: same registry name,
same constructor keywords, same ``init_weights`` policy, same parameter names/shapes (so CLIP visual
state_dicts and reference checkpoints load with ``strict=True``) and the same
``forward(x[B,3,T,H,W]) -> [B,width,T,1,1]`` contract.  The arithmetic is NOT a translation of the
reference's eager ops: the block is re-derived for a frame-major ``[B*T, N, D]`` layout with
one ``ln_1``, one fused QKV projection, a collapsed single-key cross-attention and hand-written
forward AND backward (frozen GEMMs need dgrad only) -- see DESIGN.md.

There is no eager/CPU fallback: ``forward`` raises unless the input is on a GPU and the HIP
library is built.
"""
import logging
import os
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
from torch import nn

from . import ops
from .registry import BACKBONES

BF16, F32 = torch.bfloat16, torch.float32
_LOG = logging.getLogger("aim_amd")


def get_root_logger():
    return _LOG


# ----------------------------------------------------------------------------------------------
# parameter containers (names follow the reference so state_dicts interchange)
# ----------------------------------------------------------------------------------------------
class Adapter(nn.Module):
    """Bottleneck adapter parameters (reference ``Adapter``, vit_clip.py:51-69)."""

    def __init__(self, D_features: int, mlp_ratio: float = 0.25, skip_connect: bool = True):
        super().__init__()
        self.skip_connect = skip_connect
        hidden = int(D_features * mlp_ratio)
        self.D_fc1 = nn.Linear(D_features, hidden)
        self.D_fc2 = nn.Linear(hidden, D_features)


class LayerNorm(nn.LayerNorm):
    """Parameter container; the fp32 LayerNorm itself runs in ``aim_layernorm_fwd`` (vit_clip.py:71-77)."""


class QuickGELU(nn.Module):
    """Placeholder so ``mlp`` keeps the reference's three-entry Sequential (vit_clip.py:80-82,93-97)."""


class ResidualAttentionBlock(nn.Module):
    """Parameters of one block (reference vit_clip.py:85-118); compute lives in ``_BackboneFn``."""

    def __init__(self, d_model: int, n_head: int, scale: float = 1., num_frames: int = 8, drop_path: float = 0.):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)   # parameter container only, as in the reference
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, d_model * 4)),
            ("gelu", QuickGELU()),
            ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.n_head = n_head
        self.d_model = d_model
        self.MLP_Adapter = Adapter(d_model, skip_connect=False)
        self.S_Adapter = Adapter(d_model, skip_connect=False)
        self.scale = scale
        self.T_Adapter = Adapter(d_model, skip_connect=False)
        self.num_frames = num_frames
        self.drop_prob = float(drop_path)


class Transformer(nn.Module):
    def __init__(self, num_frames, width, layers, heads, scale=1., drop_path=0.1):
        super().__init__()
        self.width, self.layers = width, layers
        dpr = [x.item() for x in torch.linspace(0, drop_path, layers)]   # vit_clip.py:297
        self.resblocks = nn.Sequential(*[
            ResidualAttentionBlock(width, heads, scale, num_frames, dpr[i]) for i in range(layers)])


_ADAPTERS = ("MLP_Adapter", "S_Adapter", "T_Adapter")
_ADAPTER_LEAVES = ("D_fc1.weight", "D_fc1.bias", "D_fc2.weight", "D_fc2.bias")


# ----------------------------------------------------------------------------------------------
# bf16 operand staging
# ----------------------------------------------------------------------------------------------
def _cast(w: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    w2 = w.detach().reshape(w.shape[0], -1).contiguous().float()
    R, C = w2.shape
    out = torch.empty((C, R) if transpose else (R, C), dtype=BF16, device=w.device)
    ops.cast_bf16(w2, out, transpose)
    return out


class _Frozen:
    """bf16 copies (and transposes, for dgrad) of one block's frozen weights; built once."""

    def __init__(self, blk: ResidualAttentionBlock):
        a = blk.attn
        self.Wqkv, self.WqkvT = _cast(a.in_proj_weight), _cast(a.in_proj_weight, True)
        self.Wo, self.WoT = _cast(a.out_proj.weight), _cast(a.out_proj.weight, True)
        # Frozen MLP and the trainable MLP_Adapter share their GEMMs (same input xn, outputs summed): the
        # operands are concatenated once -- [W_fc ; D_fc1] along N, [W_proj | D_fc2] along K -- and only the
        # adapter slices are re-cast each step (stage_mlp_adapter).  Both orientations for fwd and dgrad.
        D = blk.d_model
        r = blk.MLP_Adapter.D_fc1.weight.shape[0]
        dev = a.in_proj_weight.device
        self.D, self.r, self.H4 = D, r, 4 * D
        self.Wcat1 = torch.empty((4 * D + r, D), dtype=BF16, device=dev)      # [W_fc ; W1]      fwd  (N-concat)
        self.Wcat2 = torch.empty((D, 4 * D + r), dtype=BF16, device=dev)      # [W_proj | W2]    fwd  (K-concat)
        self.WcatT2 = torch.empty((4 * D + r, D), dtype=BF16, device=dev)     # [W_proj^T ; W2^T] dgrad (N-concat)
        self.WcatT1 = torch.empty((D, 4 * D + r), dtype=BF16, device=dev)     # [W_fc^T | W1^T]   dgrad (K-concat)
        wfc, wpr = blk.mlp.c_fc.weight.detach().float().contiguous(), blk.mlp.c_proj.weight.detach().float().contiguous()
        ops.cast_bf16(wfc, self.Wcat1[:4 * D])
        ops.cast_bf16(wpr, self.Wcat2[:, :4 * D])
        ops.cast_bf16(wpr, self.WcatT2[:4 * D], transpose=True)
        ops.cast_bf16(wfc, self.WcatT1[:, :4 * D], transpose=True)
        self.bcat1 = torch.zeros(4 * D + r, dtype=F32, device=dev)
        self.bcat1[:4 * D] = blk.mlp.c_fc.bias.detach().float()
        # bf16 operands of the two small adapters (re-cast every step by the model's cast table)
        self.small = {a: dict(W1=torch.empty((r, D), dtype=BF16, device=dev), W1T=torch.empty((D, r), dtype=BF16, device=dev),
                              W2=torch.empty((D, r), dtype=BF16, device=dev), W2T=torch.empty((r, D), dtype=BF16, device=dev))
                      for a in ("S_Adapter", "T_Adapter")}
        f = lambda p: p.detach().float().contiguous()
        self.bqkv, self.bo = f(a.in_proj_bias), f(a.out_proj.bias)
        self.bpr = f(blk.mlp.c_proj.bias)
        self.g1, self.b1 = f(blk.ln_1.weight), f(blk.ln_1.bias)
        self.g2, self.b2 = f(blk.ln_2.weight), f(blk.ln_2.bias)


    def cast_entries(self, name, w1, w2, b1=None):
        """(src, dst, mode) entries that stage one adapter's weights as bf16 operands, both orientations (mode False / True =
        plain / transposed cast; 2 = fp32 copy: the MLP_Adapter's D_fc1 bias into the concatenated bias vector)."""
        H4 = self.H4
        if name == "MLP_Adapter":      # adapter slices of the concatenated MLP operands
            ent = [(w1, self.Wcat1[H4:], False), (w2, self.Wcat2[:, H4:], False),
                   (w2, self.WcatT2[H4:], True), (w1, self.WcatT1[:, H4:], True)]
            if b1 is not None:
                ent.append((b1.reshape(1, -1), self.bcat1[H4:].reshape(1, -1), 2))
            return ent
        b = self.small[name]
        return [(w1, b["W1"], False), (w1, b["W1T"], True), (w2, b["W2"], False), (w2, b["W2T"], True)]

    def stage_mlp_adapter(self, w1, b1, w2, b2):
        """Stand-alone staging of the MLP_Adapter slices (tests / callers without the model's cast table)."""
        for src, dst, tr in self.cast_entries("MLP_Adapter", w1.detach().float().contiguous(),
                                              w2.detach().float().contiguous()):
            ops.cast_bf16(src, dst, transpose=tr)
        self.stage_mlp_bias(b1, b2)

    def stage_mlp_bias(self, b1, b2, copy_b1: bool = True):
        if copy_b1:            # (the model's cast table copies it in its one launch instead)
            self.bcat1[self.H4:] = b1.detach().float()
        self.b2row = b2.detach().float().reshape(1, -1).contiguous()


class _Frozen8:
    """fp8 e4m3 operands of one block for the inference path (BASELINE configs[4]): every large GEMM weight is
    quantised once per output channel (``ops.quantize_fp8_rows``); the MLP and its adapter keep their concatenated
    form ([W_fc ; D_fc1] along N, [W_proj | D_fc2] along K).  Built when a weight changes, not per step."""

    def __init__(self, blk: ResidualAttentionBlock):
        a, ma = blk.attn, blk.MLP_Adapter
        q = ops.quantize_fp8_rows
        self.Wqkv, self.sqkv = q(a.in_proj_weight)
        self.Wo, self.so = q(a.out_proj.weight)
        self.Wcat1, self.s1 = q(torch.cat([blk.mlp.c_fc.weight.detach().float(), ma.D_fc1.weight.detach().float()], 0))
        self.Wcat2, self.s2 = q(torch.cat([blk.mlp.c_proj.weight.detach().float(), ma.D_fc2.weight.detach().float()], 1))


class _AdapterW:
    """bf16 operands of one adapter for this step (weights are trainable: re-cast when they change).
    ``bufs`` = already staged persistent operand buffers (the model's cast table filled them)."""

    def __init__(self, w1, b1, w2, b2, bufs=None):
        if bufs is None:
            self.W1, self.W1T = _cast(w1), _cast(w1, True)     # [r, D], [D, r]
            self.W2, self.W2T = _cast(w2), _cast(w2, True)     # [D, r], [r, D]
        else:
            self.W1, self.W1T, self.W2, self.W2T = bufs["W1"], bufs["W1T"], bufs["W2"], bufs["W2T"]
        self.b1, self.b2 = b1.detach().float().contiguous(), b2.detach().float().contiguous()


def _empty(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


class _Arena:
    """Memory for the tensors that the SIDE stream's kernels produce, allocated on the MAIN stream before the fork.

    PyTorch's caching allocator ties a block to the stream that was current when it was allocated: a tensor created
    inside ``torch.cuda.stream(side)`` and later read on the main stream would go back to the side stream's pool when
    freed, where a new side-stream allocation could overwrite it while the main stream still reads it (only event
    ordering, not the allocator, would protect it).  Everything the class-token chain allocates is small (B*T rows), so
    each block carves those tensors out of ONE main-stream buffer instead; the buffer lives as long as any view of it
    (the saved context keeps views until the block's backward)."""

    def __init__(self, dev, nbytes: int):
        self.dev, self.chunk = dev, int(nbytes)
        self.buf = torch.empty(self.chunk, dtype=torch.uint8, device=dev)      # current stream == main stream here
        self.off = 0
        self.main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None

    def take(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = (n * _ELEM[dtype] + 255) // 256 * 256
        if self.off + nbytes > self.buf.numel():           # rare: grow, still on the main stream's pool
            with torch.cuda.stream(self.main):
                self.buf = torch.empty(max(self.chunk, nbytes), dtype=torch.uint8, device=self.dev)
            self.off = 0
        v = self.buf[self.off:self.off + nbytes].view(dtype)[:n].view(shape)
        self.off += nbytes
        return v


_ELEM = {BF16: 2, F32: 4}


# The class-token path of a block (temporal attention, T_/S_Adapter, the cross term: a dozen GEMMs on
# B*T = 512 rows) occupies a few CUs for ~100 us.  It runs on a second HIP stream beside the spatial
# attention kernels, which do not depend on it; the two are joined with events before `lamda`.
_SIDE = {}
_DETACHED = {}

_JOIN_STATS = [] if os.environ.get("AIM_JOIN_STATS") else None     # (tag, event, event) per join


def join_stats():
    """ms the main stream spent waiting in _Fork.join(), per tag (AIM_JOIN_STATS=1; call after a device sync)."""
    out = {}
    for tag, e0, e1 in _JOIN_STATS or []:
        out[tag] = out.get(tag, 0.0) + e0.elapsed_time(e1)
    if _JOIN_STATS is not None:
        _JOIN_STATS.clear()
    return out
_USE_SIDE = os.environ.get("AIM_SIDE_STREAM", "1") != "0"
_LAMBDA_ON_SIDE = os.environ.get("AIM_LAMBDA_SIDE", "1") != "0"
_LATE_JOIN = os.environ.get("AIM_LATE_JOIN", "1") != "0"
_FSUM_IN_LN = os.environ.get("AIM_FSUM_IN_LN", "1") != "0"     # (A/B switch) per-frame d(x1) sums from the ln_2 backward
_CLS_EARLY = os.environ.get("AIM_CLS_EARLY", "1") != "0"
_LAMBDA_FUSED = os.environ.get("AIM_LAMBDA_FUSED", "1") != "0"
_DETACH_WGRAD = os.environ.get("AIM_DETACH_WGRAD", "1") != "0"
_DETACH_BIG = os.environ.get("AIM_DETACH_BIG", "1") != "0"
_EXPSUM_DETACHED = os.environ.get("AIM_EXPSUM_DETACHED", "0") != "0"      # measured: -0.7 % (the GEMM doubles beside the attention)
_DETACHED_PRIORITY = int(os.environ.get("AIM_DETACHED_PRIORITY", "0"))   # HIP stream priority of the weight-gradient stream
# The buffers the activations' backward reads (`hcat_pre`, the adapters' `pre`) hold the activation's DERIVATIVE at the
# pre-activation, written by the forward epilogue, instead of the pre-activation itself (aim_gemm_args.aux_grad): the dgrad
# epilogues lose their exp / rcp.  AIM_AUX_GRAD=0 restores the pre-activation form (A/B runs).
_AUX_GRAD = os.environ.get("AIM_AUX_GRAD", "1") != "0"
# ... and the MLP's in the GEMM pair's own fragment order (aux_frag; large-tile kernel only, so not under AIM_GEMM_TILE=128)
_AUX_FRAG = os.environ.get("AIM_AUX_FRAG", "1") != "0" and os.environ.get("AIM_GEMM_TILE", "") != "128"
# CUs every large backward GEMM leaves out of its persistent grid (data-parallel runs: room for RCCL's kernels to start beside
# them).  Default 0: the backward already opens a hole in every shader engine once per layer (the attention backward's 224
# workgroups) and the LayerNorm backward kernels are short workgroups that free CUs continuously, while a reservation costs
# every N = 768 GEMM a whole tile round (1 182 tiles: 5 rounds on 256 CUs, 6 on 224: +20 %).  Unmeasured on > 1 GPU.
_DP_RESERVE = int(os.environ.get("AIM_DP_RESERVE_CUS", "0"))
_FP8_RES16 = os.environ.get("AIM_FP8_RES16", "1") != "0"      # fp8 inference: bf16 residual stream (0: fp32, the A/B form)
_EXPSUM_BORDER = os.environ.get("AIM_EXPSUM_BORDER", "1") != "0"      # N = 257: one 256 x 256 tile per frame + aim_qk_border
_QKV_RESERVE = int(os.environ.get("AIM_QKV_RESERVE", "32"))     # CUs the forward QKV GEMM leaves to the class-token chain (measured: 0/8/16 equal, 32 +0.7 %, 48 equal)


class _Fork:
    """Side-stream sections beside the current (main) stream.

    ``with f.side():`` queues its body on the side stream, ordered after everything queued on the main stream when
    the section is entered (the first section) or when ``f.sync_side_to_main()`` was last called.  ``f.join()``
    makes the main stream wait for all sections.  With ``AIM_SIDE_STREAM=0`` every section simply runs inline."""

    def __init__(self, dev, tag: str = ""):
        self.enabled = _USE_SIDE
        self.dev = dev
        self.tag = tag
        self.started = False
        if self.enabled:
            key = (dev.type, dev.index)
            if key not in _SIDE:
                # high priority: the side stream carries a dozen tiny kernels that the main stream later waits for
                _SIDE[key] = torch.cuda.Stream(device=dev, priority=int(os.environ.get("AIM_SIDE_PRIORITY", "-1")))
            self.side_stream = _SIDE[key]
            self.main = torch.cuda.current_stream(dev)

    def sync_main_to_side(self):
        """The main stream waits for everything queued on the side stream so far."""
        if self.enabled and self.started:
            ev = torch.cuda.Event()
            ev.record(self.side_stream)
            if _JOIN_STATS is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.main)
                self.main.wait_event(ev)
                e1.record(self.main)
                _JOIN_STATS.append((self.tag + "-mid", e0, e1))
                return
            self.main.wait_event(ev)

    def sync_side_to_main(self):
        """The side stream waits for everything queued on the main stream so far."""
        if self.enabled:
            ev = torch.cuda.Event()
            ev.record(self.main)
            self.side_stream.wait_event(ev)

    def side(self):
        fork = self

        class _Section:
            def __enter__(self_inner):
                if fork.enabled:
                    if not fork.started:
                        fork.started = True
                        fork.sync_side_to_main()
                    self_inner.ctx = torch.cuda.stream(fork.side_stream)
                    self_inner.ctx.__enter__()
                    if _JOIN_STATS is not None:      # diagnostics: the section's own span on the side stream
                        self_inner.t0 = torch.cuda.Event(enable_timing=True)
                        self_inner.t0.record(fork.side_stream)
                return fork

            def __exit__(self_inner, *exc):
                if fork.enabled:
                    if _JOIN_STATS is not None:
                        t1 = torch.cuda.Event(enable_timing=True)
                        t1.record(fork.side_stream)
                        _JOIN_STATS.append((fork.tag + "-span", self_inner.t0, t1))
                    self_inner.ctx.__exit__(*exc)
                return False

        return _Section()

    # ``with _Fork(dev) as f:`` = one side section
    def __enter__(self):
        self._sec = self.side()
        return self._sec.__enter__()

    def __exit__(self, *exc):
        return self._sec.__exit__(*exc)

    def run_detached(self, calls: list, keep: list):
        """Run ``calls`` (closures launching kernels) after the side stream's work so far, on a third stream that nothing
        waits for until ``join_detached``.  ``keep`` receives the closures so that the tensors they captured (allocated
        on other streams) outlive their use."""
        if not calls:
            return
        if not (self.enabled and _DETACH_WGRAD):
            ctx = self.side() if self.enabled else None
            if ctx is not None:
                with ctx:
                    for f in calls:
                        f()
            else:
                for f in calls:
                    f()
            return
        key = (self.dev.type, self.dev.index)
        if key not in _DETACHED:
            _DETACHED[key] = torch.cuda.Stream(device=self.dev, priority=_DETACHED_PRIORITY)
        w = _DETACHED[key]
        ev = torch.cuda.Event()
        ev.record(self.side_stream)
        w.wait_event(ev)
        with torch.cuda.stream(w):
            for f in calls:
                f()
        keep.extend(calls)

    def run_beside(self, fn):
        """Run ``fn`` on the third stream, ordered after the main stream's work so far; returns the event that marks
        its completion (None when streams are off: ``fn`` then ran inline)."""
        if not self.enabled:
            fn()
            return None
        key = (self.dev.type, self.dev.index)
        if key not in _DETACHED:
            _DETACHED[key] = torch.cuda.Stream(device=self.dev, priority=_DETACHED_PRIORITY)
        w = _DETACHED[key]
        ev = torch.cuda.Event()
        ev.record(self.main)
        w.wait_event(ev)
        with torch.cuda.stream(w):
            fn()
        done = torch.cuda.Event()
        done.record(w)
        return done

    @staticmethod
    def streams(dev):
        """The streams of this device that may carry gradient-producing kernels (current, side, detached)."""
        key = (dev.type, dev.index)
        out = [torch.cuda.current_stream(dev)]
        for table in (_SIDE, _DETACHED):
            if key in table:
                out.append(table[key])
        return out

    @staticmethod
    def join_detached(dev):
        """The current stream waits for everything queued by ``run_detached``."""
        w = _DETACHED.get((dev.type, dev.index))
        if w is not None:
            ev = torch.cuda.Event()
            ev.record(w)
            main = torch.cuda.current_stream(dev)
            if _JOIN_STATS is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                main.wait_event(ev)
                e1.record(main)
                _JOIN_STATS.append(("detached", e0, e1))
                return
            main.wait_event(ev)

    def join(self):
        if self.enabled and self.started:
            done = torch.cuda.Event()
            done.record(self.side_stream)
            if _JOIN_STATS is not None:      # diagnostics: how long the main stream sits in this wait
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.main)
                self.main.wait_event(done)
                e1.record(self.main)
                _JOIN_STATS.append((self.tag, e0, e1))
                return
            self.main.wait_event(done)


# ----------------------------------------------------------------------------------------------
# one block: forward / backward on raw buffers
# ----------------------------------------------------------------------------------------------
def _adapter_fwd_small(x_bf, ad: _AdapterW, rows, r, D, ar: _Arena, out_f32: bool):
    """Adapter on a few rows (class-token / per-frame vectors): D_fc1 -> GELU(erf) -> D_fc2."""
    pre, h = ar.take((rows, r), BF16), ar.take((rows, r), BF16)
    ops.gemm(x_bf, ad.W1, ops.EPI_ACT, h, bias=ad.b1, out2=pre, act=ops.ACT_GELU, aux_grad=_AUX_GRAD)
    out = ar.take((rows, D), F32 if out_f32 else BF16)
    ops.gemm(h, ad.W2, ops.EPI_F32 if out_f32 else ops.EPI_BF16, out, bias=ad.b2)
    return out, pre, h


def _block_forward(x, fz: _Frozen, adp: Dict[str, _AdapterW], B, T, N, H, dms1, dms2, save: bool, f8: Optional[_Frozen8] = None):
    """x: [B*T*N, D] f32 -> x2 (same shape).  Returns (x2, ctx) with ctx the tensors backward needs.

    ``f8`` (inference only, ``save`` must be False): the four large GEMMs of the block run on fp8 e4m3 operands
    (``aim_gemm_fp8``); their A operands are written as fp8 by the producing kernel (LayerNorm, attention, the FC
    epilogue).  The class-token chain (B*T rows) and the lamda statistics stay on the bf16 kernels."""
    assert f8 is None or not save
    dev = x.device
    M, D = x.shape
    BT = B * T
    r = fz.r
    # Class-token path on the side stream: temporal attention over the class tokens + T_Adapter (vit_clip.py:220-229),
    # then the cross term.  It needs q/k/v of the B*T class rows only, so those are projected apart (ln_1 + QKV on
    # 512 rows) and the chain starts beside the big ln_1 / QKV GEMM instead of after them (AIM_CLS_EARLY=0: after).
    fork = _Fork(dev, "fwd")
    # every tensor a side-stream kernel writes comes out of this main-stream buffer (see _Arena)
    ar = _Arena(dev, 48 * BT * D + 4 * B * H * T * T + 8 * BT * N + (1 << 16))

    def cls_chain(qkv_src, n_stride):
        ot = ar.take((BT, D), BF16)
        probs = ar.take((B, H, T, T), F32)
        ops.cls_attn_fwd(qkv_src, ot, probs, B, T, n_stride, H)
        ta = ar.take((BT, D), BF16)
        ops.gemm(ot, fz.Wo, ops.EPI_BF16, ta, bias=fz.bo)
        xt, t_pre, t_h = _adapter_fwd_small(ta, adp["T_Adapter"], BT, r, D, ar, out_f32=False)
        # cross-attention to the single key/value xt[bt] (:265): softmax == 1, so crs = out_proj(W_v xt + b_v)
        kv = ar.take((BT, 2 * D), BF16)
        ops.gemm(xt, fz.Wqkv[D:], ops.EPI_BF16, kv, bias=fz.bqkv[D:])
        crs = ar.take((BT, D), F32)
        ops.gemm(kv[:, D:], fz.Wo, ops.EPI_F32, crs, bias=fz.bo)
        return probs, ta, t_pre, t_h, kv, crs

    if _CLS_EARLY:
        with fork.side():
            xl_cls = ar.take((BT, D), BF16)
            if x.dtype == BF16:      # (the fp8 path's residual stream is bf16)
                ops.layernorm_fwd_x16(x, fz.g1, fz.b1, BT, D, N * D, y_bf16=xl_cls)
            else:
                ops.layernorm_fwd(x, fz.g1, fz.b1, BT, D, N * D, y_bf16=xl_cls, mean=ar.take((BT,), F32),
                                  rstd=ar.take((BT,), F32))
            qkv_cls = ar.take((BT, 3 * D), BF16)
            ops.gemm(xl_cls, fz.Wqkv, ops.EPI_BF16, qkv_cls, bias=fz.bqkv)
            probs, ta, t_pre, t_h, kv, crs = cls_chain(qkv_cls, 1)      # "N = 1": the rows ARE the class tokens
    # main stream: ln_1 (once) + fused QKV projection
    qkv = _empty((M, 3 * D), BF16, dev)
    reserve = _QKV_RESERVE if (_CLS_EARLY and fork.enabled) else 0     # CUs left to the class-token chain (per-call argument)
    if f8 is not None:
        xl = _empty((M, D), ops.FP8, dev)
        mean1 = rstd1 = None
        if x.dtype == BF16:
            ops.layernorm_fwd_x16(x, fz.g1, fz.b1, M, D, D, y8=xl)
        else:
            ops.layernorm_fwd_fp8(x, fz.g1, fz.b1, M, D, D, xl)
        ops.gemm_fp8(xl, f8.Wqkv, f8.sqkv, ops.EPI_BF16, qkv, bias=fz.bqkv, reserve_cus=reserve)
    else:
        xl = _empty((M, D), BF16, dev)
        mean1, rstd1 = _empty((M,), F32, dev), _empty((M,), F32, dev)
        ops.layernorm_fwd(x, fz.g1, fz.b1, M, D, D, y_bf16=xl, mean=mean1, rstd=rstd1)
        ops.gemm(xl, fz.Wqkv, ops.EPI_BF16, qkv, bias=fz.bqkv, reserve_cus=reserve)
    del xl
    if not _CLS_EARLY:
        with fork.side():
            probs, ta, t_pre, t_h, kv, crs = cls_chain(qkv, N)
    # lamda = cw / (cw + ow)  (:149-151,184-186,272; no grad): `ow` is a batched 197x197x768 GEMM reduced to (max, sum exp)
    # partials, independent of the class-token path and of the spatial attention (:264).  Fused form: kx, ready early on
    # the side stream, rides that GEMM as a 198th key, so `cw` comes out of the same launch (no separate pass over q).
    fused = _LAMBDA_FUSED and _CLS_EARLY and N < 256 and ops.expsum_tiles(N, N) == 8
    ss, part_ready = None, None
    if fused:
        fork.sync_main_to_side()          # kx
        nt = 16
        part = _empty((BT, nt, 2), F32, dev)
        ops.gemm(qkv, qkv[:, D:], ops.EPI_EXPSUM, part, M=N, N=N, K=D, batch=BT, stride_a=N * 3 * D,
                 stride_w=N * 3 * D, scale=0.125, xrow=kv)
    else:
        # N = 257 (ViT-L/14): the scores of tokens 0..255 against each other are ONE full tile per frame of the persistent
        # 256 x 256 kernel (8 slots); the 257th row and column and the cross scores come from one pass over q and k
        # (aim_qk_border: slots 8, 9).  Other N > 256 or N <= 128: the 128 x 128 kernel's tiles + qk_cross.
        border = _EXPSUM_BORDER and N == 257 and D in (512, 1024) and _LAMBDA_ON_SIDE and _CLS_EARLY
        nt = 10 if border else ops.expsum_tiles(N, N)
        part = _empty((BT, nt, 2), F32, dev)
        # the cross scores q_i . kx (one pass over q) go to the side stream as soon as q exists, beside the ow GEMM
        if _LAMBDA_ON_SIDE and _CLS_EARLY:
            fork.sync_side_to_main()
            with fork.side():
                ss = ar.take((BT, N), F32)
                if border:
                    ops.qk_border(qkv, kv, ss, part, 8, BT, N, D, 0.125)
                else:
                    ops.qk_cross(qkv, kv, ss, BT, N, D, 0.125)

        def expsum():
            if border:
                ops.gemm(qkv, qkv[:, D:], ops.EPI_EXPSUM, part, M=N - 1, N=N - 1, K=D, batch=BT, stride_a=N * 3 * D,
                         stride_w=N * 3 * D, scale=0.125, slot_stride=2 * nt)
            else:
                ops.gemm(qkv, qkv[:, D:], ops.EPI_EXPSUM, part, M=N, N=N, K=D, batch=BT, stride_a=N * 3 * D,
                         stride_w=N * 3 * D, scale=0.125)

        part_ready = fork.run_beside(expsum) if _EXPSUM_DETACHED else expsum()

    def lamda_chain():
        lam, oml = ar.take((BT,), F32), ar.take((BT,), F32)
        if fused:
            ops.lambda_partials(part, lam, oml, BT)
        else:
            ops.lambda_(qkv, kv, part, nt, lam, oml, BT, N, D, 0.125, ss=ss)
        # S_Adapter(lamda * crs_attn): a per-frame vector broadcast over tokens (:275)
        sin = ar.take((BT, D), BF16)
        ops.scale_rows(crs, lam, y=sin)
        return (lam, oml, sin) + _adapter_fwd_small(sin, adp["S_Adapter"], BT, r, D, ar, out_f32=True)

    if _LAMBDA_ON_SIDE:               # the chain runs beside the spatial attention instead of after it
        if part_ready is not None:
            fork.side_stream.wait_event(part_ready)      # it needs `part`
        else:
            fork.sync_side_to_main()
        with fork.side():
            lam, oml, sin, sv, s_pre, s_h = lamda_chain()
    if f8 is not None:
        ao, lse = _empty((M, D), ops.FP8, dev), None
        ops.attn_fwd_fp8(qkv, ao, BT, N, H)
    else:
        ao = _empty((M, D), BF16, dev)
        lse = _empty((BT, H, N), F32, dev)
        ops.attn_fwd(qkv, ao, lse, BT, N, H)
    fork.join()
    if not _LAMBDA_ON_SIDE:
        if part_ready is not None:
            torch.cuda.current_stream(dev).wait_event(part_ready)
        lam, oml, sin, sv, s_pre, s_h = lamda_chain()
    # x1 = x + (1 - lamda) * out_proj(ao) + drop_path(scale * s_vec)
    H4 = 4 * D
    if f8 is not None:
        # the residual stream of the fp8 path is bf16 (AIM_EPI_RES16: resid + update rounded once per update): against fp8
        # GEMM operands (2^-4 relative) a 2^-9 rounding of the stream is noise, and the two residual GEMMs of a block move half
        # the bytes of the fp32 form (they are store-bound at K = 1024)
        r16 = x.dtype == BF16
        epi_res, rdt = (ops.EPI_RES16, BF16) if r16 else (ops.EPI_F32, F32)
        x1 = _empty((M, D), rdt, dev)
        ops.gemm_fp8(ao, f8.Wo, f8.so, epi_res, x1, bias=fz.bo, resid=x, af=oml, vec=sv, bt=dms1, ntok=N)
        xn = _empty((M, D), ops.FP8, dev)
        if r16:
            ops.layernorm_fwd_x16(x1, fz.g2, fz.b2, M, D, D, y8=xn)
        else:
            ops.layernorm_fwd_fp8(x1, fz.g2, fz.b2, M, D, D, xn)
        hcat = _empty((M, H4 + r), ops.FP8, dev)
        ops.gemm_fp8(xn, f8.Wcat1, f8.s1, ops.EPI_ACT8, hcat, bias=fz.bcat1, act=ops.ACT_QGELU, n_split=H4,
                     act2=ops.ACT_GELU, at=dms2, ntok=N)
        x2 = _empty((M, D), rdt, dev)
        ops.gemm_fp8(hcat, f8.Wcat2, f8.s2, epi_res, x2, bias=fz.bpr, resid=x1, vec=fz.b2row, ldv=0, bt=dms2, ntok=N)
        return x2, None
    x1 = _empty((M, D), F32, dev)
    ops.gemm(ao, fz.Wo, ops.EPI_F32, x1, bias=fz.bo, resid=x, af=oml, vec=sv, bt=dms1, ntok=N)
    x2, xn, mean2, rstd2, hcat_pre, a_s = _mlp_adapter_forward(x1, fz, dms2, N, save)
    ctx = None
    if save:
        ctx = dict(x=x, mean1=mean1, rstd1=rstd1, qkv=qkv, probs=probs, ta=ta, t_pre=t_pre, t_h=t_h, lam=lam,
                   oml=oml, ao=ao, lse=lse, sin=sin, s_pre=s_pre, s_h=s_h, x1=x1, mean2=mean2, rstd2=rstd2, xn=xn,
                   hcat_pre=hcat_pre, a_s=a_s, dms1=dms1, dms2=dms2)
    return x2, ctx


def _mlp_adapter_forward(x1, fz: _Frozen, dms2, N, save: bool):
    """Joint adaptation (vit_clip.py:285-286; identical in vitclip_aim.py:209-210): x2 = x1 + mlp(ln_2(x1)) +
    drop_path(scale * MLP_Adapter(ln_2(x1))).  One GEMM for [c_fc | D_fc1] (N = 4D + r; QuickGELU on the MLP columns,
    dms2 * GELU on the adapter's) and one for [c_proj | D_fc2] (K = 4D + r); the adapter's token-scaled bias rides along
    as `vec`.  Returns (x2, xn, mean2, rstd2, hcat_pre, a_s)."""
    dev = x1.device
    M, D = x1.shape
    r, H4 = fz.r, 4 * D
    xn = _empty((M, D), BF16, dev)
    mean2, rstd2 = _empty((M,), F32, dev), _empty((M,), F32, dev)
    ops.layernorm_fwd(x1, fz.g2, fz.b2, M, D, D, y_bf16=xn, mean=mean2, rstd=rstd2)
    # the pre-activation is only needed by a backward: a no-grad forward passes no `out2` (the epilogue's stores to an empty
    # buffer resource are dropped: 658 MB per ViT-B block less to write)
    frag = _AUX_FRAG and M >= 1024          # the large-tile kernel pair keeps it in its own fragment order (no re-tiling)
    if frag:
        hcat_pre = ops.frag_buffer(M, H4 + r, dev) if save else None
    else:
        hcat_pre = _empty((M, H4 + r), BF16, dev) if (save or M < 1024) else None     # (the small-M kernel always stores it)
    hcat = _empty((M, H4 + r), BF16, dev)
    ops.gemm(xn, fz.Wcat1, ops.EPI_ACT, hcat, bias=fz.bcat1, out2=hcat_pre, act=ops.ACT_QGELU, n_split=H4,
             act2=ops.ACT_GELU, at=dms2, ntok=N, aux_grad=_AUX_GRAD, aux_frag=frag and hcat_pre is not None)
    x2 = _empty((M, D), F32, dev)
    ops.gemm(hcat, fz.Wcat2, ops.EPI_F32, x2, bias=fz.bpr, resid=x1, vec=fz.b2row, ldv=0, bt=dms2, ntok=N)
    # the adapter's activation slice stays a VIEW of hcat (wgrad takes a row stride): no copy kernel in the forward, at the
    # price of keeping hcat (M x (4D + r) bf16) alive until this block's backward
    a_s = hcat[:, H4:] if save else None
    return x2, xn, mean2, rstd2, hcat_pre, a_s


def _mlp_adapter_backward(dyb, x_in, mean2, rstd2, xn, hcat_pre, a_s, dms2, fz: _Frozen, gm, N, fsum=None):
    """Backward of ``_mlp_adapter_forward``: returns (d(x_in) as bf16, the weight-gradient closures).  x2 = x_in +
    [h | a_s] [W_proj | W2]^T + b_proj + dms2[tok] * b2.  ``fsum = (w [N], partial [frames, LN_FSUM_GROUPS, D])``: the ln_2
    backward also leaves the per-frame sums of w[n] * d(x_in) (in token groups) in ``partial``."""
    dev = dyb.device
    M, D = dyb.shape
    r, H4 = fz.r, 4 * D
    big_later: list = []
    # D_fc2: weight gradient + the DropPath-scaled bias gradient in one pass over dyb
    if _DETACH_BIG:
        big_later.append(lambda: ops.wgrad(dyb, a_s, gm["D_fc2.weight"], gm["D_fc2.bias"], at=dms2, ntok=N))
    else:
        ops.wgrad(dyb, a_s, gm["D_fc2.weight"], gm["D_fc2.bias"], at=dms2, ntok=N)
    dcat = _empty((M, H4 + r), BF16, dev)           # [dh_pre | da_pre]
    ops.gemm(dyb, fz.WcatT2, ops.EPI_DACT, dcat, aux=hcat_pre, act=ops.ACT_QGELU, n_split=H4, act2=ops.ACT_GELU,
             at=dms2, ntok=N, aux_grad=_AUX_GRAD, aux_frag=_AUX_FRAG and dyb.shape[0] >= 1024, reserve_cus=_DP_RESERVE)
    if _DETACH_BIG:
        big_later.append(lambda: ops.wgrad(dcat[:, H4:], xn, gm["D_fc1.weight"], gm["D_fc1.bias"]))
    else:
        ops.wgrad(dcat[:, H4:], xn, gm["D_fc1.weight"], gm["D_fc1.bias"])
    dxn = _empty((M, D), BF16, dev)
    ops.gemm(dcat, fz.WcatT1, ops.EPI_BF16, dxn, reserve_cus=_DP_RESERVE)     # K = 4D + r: frozen c_fc dgrad + adapter D_fc1 dgrad
    dxb = _empty((M, D), BF16, dev)                  # (dcat stays alive in the D_fc1 weight-gradient closure)
    if fsum is not None:
        ops.layernorm_bwd_fsum(dxn, x_in, fz.g2, mean2, rstd2, dyb, dxb, fsum[0], fsum[1], M // N, N, D)
    else:
        ops.layernorm_bwd(dxn, x_in, fz.g2, mean2, rstd2, M, D, lddy=D, ldx=D, lddx=D, dres=dyb, dx_bf16=dxb)
    return dxb, big_later


def _adapter_bwd_small(dout_bf, ad: _AdapterW, a_in, pre, h, grads, rows, r, D, ar: _Arena, need_dx_bf16: bool, later: list):
    """Backward of an adapter on a few rows; returns d(input).  Its 4 parameter gradients are not on the gradient
    path: the two wgrad calls are appended to ``later`` (run by the caller off the critical stream)."""
    later.append(lambda: ops.wgrad(dout_bf, h, grads["D_fc2.weight"], grads["D_fc2.bias"]))
    dpre = ar.take((rows, r), BF16)
    ops.gemm(dout_bf, ad.W2T, ops.EPI_DACT, dpre, aux=pre, act=ops.ACT_GELU, aux_grad=_AUX_GRAD)
    later.append(lambda: ops.wgrad(dpre, a_in, grads["D_fc1.weight"], grads["D_fc1.bias"]))
    din = ar.take((rows, D), BF16 if need_dx_bf16 else F32)
    ops.gemm(dpre, ad.W1T, ops.EPI_BF16 if need_dx_bf16 else ops.EPI_F32, din)
    return din


def _block_backward(dyb, c, fz: _Frozen, adp: Dict[str, _AdapterW], grads, B, T, N, H, keep: Optional[list] = None):
    """dyb = d(loss)/d(x2) [M, D] -> d(loss)/d(x); adapter grads accumulated into ``grads``.

    The residual-stream GRADIENT is carried in bf16 (one tensor serves as the running residual gradient and
    as the dgrad GEMMs' operand; LayerNorm backward adds in fp32 and rounds once per block).  The forward
    residual stream stays fp32.  The reference's apex-O1 run keeps these gradients in fp16."""
    dev = dyb.device
    M, D = dyb.shape
    BT = B * T
    r, H4 = fz.r, 4 * D
    # sum_n dms1[n] * d(x1)[frame, n, :] (the per-frame S_Adapter vector's gradient) comes out of the ln_2 backward itself
    fpart = _empty((BT, ops.LN_FSUM_GROUPS, D), F32, dev) if _FSUM_IN_LN else None
    dx1b, big_later = _mlp_adapter_backward(dyb, c["x1"], c["mean2"], c["rstd2"], c["xn"], c["hcat_pre"], c["a_s"], c["dms2"],
                                            fz, grads["MLP_Adapter"], N, fsum=(c["dms1"], fpart) if _FSUM_IN_LN else None)
    # ---- x1 = x + oml[f] * (ao Wo^T + bo) + dms1[tok] * s_vec[f]
    # class-token chain (S_Adapter, cross term, T_Adapter; a dozen kernels on B*T rows) on the side stream ...
    later: list = big_later       # the adapters' weight gradients: nobody downstream waits for them
    ar = _Arena(dev, 48 * BT * D + (1 << 16))      # side-stream tensors live in main-stream memory (see _Arena)
    with _Fork(dev, "bwd") as fork:
        dsv = ar.take((BT, D), F32)
        if _FSUM_IN_LN:
            ops.frame_sum(fpart, None, dsv, BT, ops.LN_FSUM_GROUPS, D)      # the groups' partial sums, in order
        else:
            ops.frame_sum(dx1b, c["dms1"], dsv, BT, N, D)
        # S_Adapter on the per-frame vector sin = lamda * crs ; crs = (xt Wv^T + bv) Wo^T + bo
        dsv_b = ar.take((BT, D), BF16)
        ops.cast_bf16(dsv, dsv_b)
        dsin = _adapter_bwd_small(dsv_b, adp["S_Adapter"], c["sin"], c["s_pre"], c["s_h"], grads["S_Adapter"], BT, r,
                                  D, ar, need_dx_bf16=False, later=later)
        dcrs = ar.take((BT, D), BF16)
        ops.scale_rows(dsin, c["lam"], y=dcrs)
        dvx = ar.take((BT, D), BF16)
        ops.gemm(dcrs, fz.WoT, ops.EPI_BF16, dvx)
        dxt = ar.take((BT, D), BF16)
        ops.gemm(dvx, fz.WqkvT[:, 2 * D:], ops.EPI_BF16, dxt)
        # T_Adapter and out_proj of the temporal attention over class tokens
        dta = _adapter_bwd_small(dxt, adp["T_Adapter"], c["ta"], c["t_pre"], c["t_h"], grads["T_Adapter"], BT, r, D,
                                 ar, need_dx_bf16=True, later=later)
        dot = ar.take((BT, D), BF16)
        ops.gemm(dta, fz.WoT, ops.EPI_BF16, dot)
        # the class rows' share of d(qkv) and its QKV dgrad stay on the side stream: the main stream's big dgrad GEMM
        # below does not wait for the class-token chain
        if _LATE_JOIN:
            dqkv_cls = ar.take((BT, 3 * D), BF16)
            ops.cls_attn_bwd(c["qkv"], c["probs"], dot, dqkv_cls, B, T, N, H, compact=True)
            dxl_cls = ar.take((BT, D), F32)
            ops.gemm(dqkv_cls, fz.WqkvT, ops.EPI_F32, dxl_cls)
    if keep is None:       # stand-alone use: the weight gradients are complete when this function returns
        keep = []
        fork.run_detached(later, keep)
        _Fork.join_detached(dev)
    else:
        fork.run_detached(later, keep)
    # ... beside the spatial attention backward and the fused QKV dgrad on the main stream
    dao = _empty((M, D), BF16, dev)
    ops.gemm(dx1b, fz.WoT, ops.EPI_BF16, dao, af=c["oml"], ntok=N, reserve_cus=_DP_RESERVE)
    dqkv = _empty((M, 3 * D), BF16, dev)
    delta = _empty((BT, H, N), F32, dev)
    ops.attn_bwd(c["qkv"], c["ao"], dao, c["lse"], delta, dqkv, BT, N, H)
    del dao
    if not _LATE_JOIN:      # (A/B switch) join first and add the class rows into d(qkv) itself
        fork.join()
        ops.cls_attn_bwd(c["qkv"], c["probs"], dot, dqkv, B, T, N, H)
    dxl = _empty((M, D), BF16, dev)
    ops.gemm(dqkv, fz.WqkvT, ops.EPI_BF16, dxl, reserve_cus=_DP_RESERVE)
    del dqkv
    if _LATE_JOIN:
        fork.join()
        ops.add_rows(dxl, N * D, dxl_cls)       # class rows: rows n == 0 of every frame
    # ---- ln_1
    dxb = _empty((M, D), BF16, dev)
    ops.layernorm_bwd(dxl, c["x"], fz.g1, c["mean1"], c["rstd1"], M, D, lddy=D, ldx=D, lddx=D, dres=dx1b, dx_bf16=dxb)
    return dxb


# ----------------------------------------------------------------------------------------------
# whole backbone as one autograd node
# ----------------------------------------------------------------------------------------------
class _BackboneFn(torch.autograd.Function):
    """imgs -> [B, D, T] features.  Differentiable inputs: temporal_embedding, ln_post.{weight,bias} and
    the 12 adapter tensors of every layer (the reference's trainable set, vit_clip.py:413-415)."""

    @staticmethod
    def forward(ctx, model: "ViT_CLIP", grad_enabled: bool, imgs: torch.Tensor, *params: torch.Tensor):
        L, H = model.layers, model.heads
        B, C, T, Hh, Ww = imgs.shape
        D, p = model.width, model.patch_size
        G = Hh // p
        N = G * G + 1
        BT, M = B * T, B * T * N
        dev = imgs.device
        temporal, lnp_w, lnp_b = params[0], params[1], params[2]
        # grad mode is off inside Function.forward and needs_input_grad stays True under torch.no_grad(): the caller's
        # grad mode decides whether the per-block contexts (~2 GB per ViT-B layer at 64 clips) are kept
        need_grad = grad_enabled and any(ctx.needs_input_grad)
        frozen = model._frozen_operands()
        staged_bias = model._stage_adapters(frozen, params)   # ONE launch: all 36 adapters' weights -> bf16 operands (+ biases)
        adp = []
        for i in range(L):
            d = {}
            for j, a in enumerate(_ADAPTERS):
                k = 3 + (i * 3 + j) * 4
                if a == "MLP_Adapter":      # shares the frozen MLP's GEMMs (concatenated operands)
                    frozen["blocks"][i].stage_mlp_bias(params[k + 1], params[k + 3], copy_b1=not staged_bias)
                else:
                    d[a] = _AdapterW(params[k], params[k + 1], params[k + 2], params[k + 3],
                                     bufs=frozen["blocks"][i].small[a])
            adp.append(d)
        # patch embedding as a GEMM (conv1: kernel = stride = patch, no bias; vit_clip.py:436)
        Kp = frozen["conv"].shape[1]
        A = _empty((BT * G * G, Kp), BF16, dev)
        ops.patchify(imgs, A, B, T, Hh, Ww, p, Kp, *model._norm_now)
        tok = _empty((BT * G * G, D), BF16, dev)
        ops.gemm(A, frozen["conv"], ops.EPI_BF16, tok)
        del A
        x = _empty((M, D), F32, dev)
        mean0, rstd0 = _empty((M,), F32, dev), _empty((M,), F32, dev)
        tmp = temporal.detach().reshape(T, D).float().contiguous()
        ops.embed_ln(tok, frozen["cls"], frozen["pos"], tmp, frozen["gpre"], frozen["bpre"], x, mean0, rstd0, B, T, N, D)
        # blocks
        ctxs: List[Optional[dict]] = []
        training = model.training
        f8 = None
        if model.inference_precision == 'fp8' and not need_grad:
            if M >= 1024 and model.variant == 'vit_clip':
                f8 = model._fp8_operands()
            elif not getattr(model, "_fp8_warned", False):
                model._fp8_warned = True
                _LOG.warning("fp8 inference was requested but this forward runs bf16: %s",
                             "the stock-AIM variant has no fp8 path" if model.variant != 'vit_clip'
                             else "fewer than 1024 token rows (M = %d)" % M)
        masks = model._drop_masks(N, training, dev)          # [L, 2, N]: all layers' DropPath factors in three launches
        aim = model.variant == 'aim'
        if aim:
            from .aim_variant import aim_block_forward
        # checkpoint=True (vit_clip.py:316-320, torch.utils.checkpoint per block): the forward keeps each block's INPUT
        # only (M x D fp32) and the DropPath factors it drew; the backward re-runs the block's forward with them to rebuild
        # its context (~2 GB per ViT-B layer at 64 clips) just before that block's backward.  Bit-identical gradients
        # (same kernels, same inputs, same factors), ~1/3 more time, 1 / L of the activation memory.
        ckpt = bool(model.checkpoint) and need_grad

        def run_block(i, x_in, save):
            dms1, dms2 = masks[i, 0], masks[i, 1]
            if aim:      # stock-AIM block: its first DropPath acts on the un-scaled temporal branch (vitclip_aim.py:205)
                scale = float(model.transformer.resblocks[i].scale)
                return aim_block_forward(x_in, frozen["blocks"][i], adp[i], B, T, N, H, dms1 * (1.0 / scale), dms2, save)
            return _block_forward(x_in, frozen["blocks"][i], adp[i], B, T, N, H, dms1, dms2, save,
                                  f8=None if f8 is None else f8[i])

        if f8 is not None and _FP8_RES16:            # the fp8 path's residual stream is bf16
            x16 = _empty((M, D), BF16, dev)
            ops.cast_bf16(x, x16)
            x = x16
        for i in range(L):
            x_in = x
            x, c = run_block(i, x_in, need_grad and not ckpt)
            ctxs.append(dict(recompute=run_block, x_in=x_in) if ckpt else c)
        # ln_post on the class rows only (LayerNorm is per-row; vit_clip.py:452-453)
        gw, gb = lnp_w.detach().float().contiguous(), lnp_b.detach().float().contiguous()
        y = _empty((BT, D), F32, dev)
        meanp, rstdp = _empty((BT,), F32, dev), _empty((BT,), F32, dev)
        if x.dtype == BF16:
            ops.layernorm_fwd_x16(x, gw, gb, BT, D, N * D, y_f32=y)
        else:
            ops.layernorm_fwd(x, gw, gb, BT, D, N * D, y_f32=y, mean=meanp, rstd=rstdp)
        if need_grad:
            ctx.model, ctx.dims = model, (B, T, N, H, D, L)
            ctx.saved = dict(ctxs=ctxs, adp=adp, tok=tok, mean0=mean0, rstd0=rstd0, tmp=tmp, xL=x, gw=gw, meanp=meanp,
                             rstdp=rstdp, params=params)
        return y.reshape(B, T, D).permute(0, 2, 1)      # '(b t) d -> b d t'

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        B, T, N, H, D, L = ctx.dims
        s = ctx.saved
        BT, M = B * T, B * T * N
        dev = dout.device
        frozen = model._frozen_operands()
        params = s["params"]
        # Gradient buffers (fp32; every kernel ACCUMULATES into them).  With `grad_in_place` (set by
        # dist.build_optimizer for the flat-buffer optimizer) the kernels add straight into the existing
        # `param.grad` views of the flat gradient buffer and autograd gets None for those inputs: no
        # temporary zero-filled tensors and no 147 small accumulate kernels per step.
        grads_out: List[Optional[torch.Tensor]] = [None] * len(params)
        in_place = [False] * len(params)

        def buf(k):
            p_ = params[k]
            if (model.grad_in_place and p_.requires_grad and p_.grad is not None and p_.grad.dtype == F32
                    and p_.grad.is_contiguous() and p_.grad.device == dev):
                in_place[k] = True
                return p_.grad
            return torch.zeros_like(p_, dtype=F32)

        layer_grads = []
        for i in range(L):
            lg = {}
            for j, a in enumerate(_ADAPTERS):
                k = 3 + (i * 3 + j) * 4
                lg[a] = {}
                for e, leaf in enumerate(_ADAPTER_LEAVES):
                    g = buf(k + e)
                    lg[a][leaf] = g
                    grads_out[k + e] = g
            layer_grads.append(lg)
        dgw, dgb = buf(1), buf(2)
        dy = dout.permute(0, 2, 1).reshape(BT, D).contiguous().float()
        # ln_post backward touches the class rows only; every other row of the top gradient is zero
        dxb = torch.zeros((M, D), dtype=BF16, device=dev)
        ops.layernorm_bwd(dy, s["xL"], s["gw"], s["meanp"], s["rstdp"], BT, D, lddy=D, ldx=N * D, lddx=N * D,
                          dx_bf16=dxb, dgamma=dgw, dbeta=dgb)
        keep: list = []        # tensors the detached weight-gradient stream still reads; dropped after join_detached
        hook = model.grad_ready_hook
        blk_bwd = _block_backward
        if model.variant == 'aim':
            from .aim_variant import aim_block_backward as blk_bwd
        for i in reversed(range(L)):
            c = s["ctxs"][i]
            if "recompute" in c:        # checkpoint=True: rebuild this block's context from its saved input
                c = c["recompute"](i, c["x_in"], True)[1]
            recomputed = "recompute" in s["ctxs"][i]
            dxb = blk_bwd(dxb, c, frozen["blocks"][i], s["adp"][i], layer_grads[i], B, T, N, H, keep)
            s["ctxs"][i] = c = None
            if recomputed:
                # the detached weight-gradient stream's closures hold this block's tensors (~0.2 GB per ViT-B layer at 8
                # clips) until it is joined: with checkpointing the join is per block, so the memory goes back now (the
                # main stream is ordered behind the detached one before it can re-use the blocks)
                _Fork.join_detached(dev)
                keep.clear()
            if hook is not None:
                # every kernel that accumulates into the gradients of layers >= i has been QUEUED (adapter weight
                # gradients on the detached stream): the data-parallel optimizer may start reducing that slice of the
                # flat gradient buffer behind events on these streams, while the backward of layers < i still runs
                k0 = 3 + i * 12
                hook(i, all(in_place[k0:k0 + 12]), _Fork.streams(dev))
        dtmp = buf(0)
        ops.embed_bwd(dxb, s["tok"], frozen["cls"], frozen["pos"], s["tmp"], frozen["gpre"], s["mean0"], s["rstd0"],
                      dtmp.view(T, D), B, T, N, D)
        grads_out[0] = dtmp.view(1, T, D)
        grads_out[1], grads_out[2] = dgw, dgb
        _Fork.join_detached(dev)       # every weight gradient is in place before autograd hands them on
        keep.clear()
        for k, p_ in enumerate(params):
            if not p_.requires_grad or in_place[k]:
                grads_out[k] = None
            elif grads_out[k] is not None and grads_out[k].dtype != p_.dtype:
                grads_out[k] = grads_out[k].to(p_.dtype)
        ctx.saved = None
        return (None, None, None) + tuple(grads_out)


@BACKBONES.register_module()
class ViT_CLIP(nn.Module):
    """ViT definition in CLIP image encoder + AIM adapters (reference vit_clip.py:327-458)."""

    def __init__(self, input_resolution: int, num_frames: int, patch_size: int, width: int, layers: int, heads: int,
                 drop_path_rate, adapter_scale=0.5, pretrained=None, shift=False, checkpoint=False):
        super().__init__()
        if shift:
            # reference vit_clip.py:233-258: raises EinopsError at every supported resolution
            raise NotImplementedError("ViT_CLIP(shift=True) is dead code in the reference and is not supported")
        if width % heads != 0 or width // heads != 64:
            raise ValueError("the HIP attention kernels are built for head_dim 64 (ViT-B/16, ViT-L/14)")
        self.input_resolution = input_resolution
        self.pretrained = pretrained
        self.patch_size = patch_size
        self.width, self.layers, self.heads = width, layers, heads
        self.conv1 = nn.Conv2d(in_channels=3, out_channels=width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.num_frames = num_frames
        self.temporal_embedding = nn.Parameter(torch.zeros(1, num_frames, width))
        self.shift = shift
        self.checkpoint = checkpoint   # per-block activation recompute in the backward (vit_clip.py:316-320; _BackboneFn)
        self.transformer = Transformer(num_frames, width, layers, heads, scale=adapter_scale, drop_path=drop_path_rate)
        self.ln_post = LayerNorm(width)
        self._frozen_cache = None
        self._norm_mean = self._norm_std = None     # set by a fused GPUNormalize hook (module_hooks.py) for the NEXT forward only
        self._norm_now = (None, None)               # what this forward's patch gather applies
        self.grad_in_place = False                  # accumulate straight into param.grad (see _BackboneFn.backward)
        self.grad_ready_hook = None                 # fn(layer, in_place, streams): set by dist.FlatAdamW (overlapped all-reduce)
        self._fp8_cache = None
        self.weights_epoch = 0                      # bumped by dist.FlatAdamW.step(): trainable weights changed in place
        self.variant = 'vit_clip'                   # 'aim': the stock-AIM block (aim_variant.py)
        # inference precision of the large GEMMs: 'bf16' (default, the training kernels) or 'fp8' (BASELINE configs[4]:
        # fp8 e4m3 operands on the block-scaled MFMA; no-grad forwards only).  AIM_INFER_FP8=1 selects fp8 globally.
        self.inference_precision = 'fp8' if os.environ.get("AIM_INFER_FP8", "0") == "1" else 'bf16'
        self._cast_table = None
        # 'bf16': the product (bf16 MFMA operands, hand-written backward).  'fp32': the reference-precision verification
        # mode (fp32_path.py: forward and the same hand-written backward on fp32 kernels), held to the reference's own fp32
        # outputs and autograd gradients at 1e-5.
        self.precision = 'bf16'

    def set_precision(self, precision: str):
        """'bf16' (default) | 'fp32': arithmetic of the forward and backward.  fp32 = f32-MFMA kernels with the reference's
        fp32 arithmetic (vit_clip.py:433-458), ~1/16 of the bf16 MFMA rate: the verification mode (one stream, plain
        autograd outputs -- no flat-buffer accumulation, no overlapped all-reduce, no activation checkpointing)."""
        if precision not in ('bf16', 'fp32'):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.precision = precision
        return self

    # ---- reference API ------------------------------------------------------------------------
    def init_weights(self, pretrained=None):
        """Reference ``init_weights`` (vit_clip.py:352-423): init, optional CLIP load, zero the adapters'
        up-projections, freeze everything but temporal_embedding / ln_post / Adapters / cls_head."""
        def _init_weights(m):
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

        if pretrained:
            self.pretrained = pretrained
        if isinstance(self.pretrained, str):
            self.apply(_init_weights)
            try:
                import clip  # noqa: F401
            except ImportError as e:
                raise RuntimeError(
                    "pretrained=%r needs the OpenAI `clip` package and its downloaded weights "
                    "(reference vit_clip.py:369-372); load a CLIP visual state_dict with "
                    "load_state_dict(strict=False) instead" % (self.pretrained,)) from e
            name = "ViT-B/16" if self.layers == 12 else "ViT-L/14"
            clip_model, _ = clip.load(name, device="cpu")
            sd = clip_model.visual.state_dict()
            del clip_model
            del sd['proj']
            msg = self.load_state_dict(sd, strict=False)
            _LOG.info('Missing keys: %s', msg.missing_keys)
            _LOG.info('Unexpected keys: %s', msg.unexpected_keys)
        elif self.pretrained is None:
            self.apply(_init_weights)
        else:
            raise TypeError('pretrained must be a str or None')
        for n, m in self.transformer.named_modules():
            if n.split(".")[-1] in _ADAPTERS:
                nn.init.constant_(m.D_fc2.weight, 0)
                nn.init.constant_(m.D_fc2.bias, 0)
        for name, param in self.named_parameters():
            if ('temporal_embedding' not in name and 'ln_post' not in name and 'Adapter' not in name
                    and 'cls_head' not in name):
                param.requires_grad = False
        n_train = sum(p.numel() for p in self.parameters() if p.requires_grad)
        n_total = sum(p.numel() for p in self.parameters())
        _LOG.info('Number of total parameters: %6.2f, tunable parameters: %6.2f', n_total / 1e6, n_train / 1e6)
        self._frozen_cache = None

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed', 'temporal_embedding'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table', 'temporal_position_bias_table'}

    # ---- operand staging ------------------------------------------------------------------------
    def _param_slots(self, root, tag):
        """(module, name) of every parameter under ``root``, walked once: the per-forward cache keys below read the CURRENT
        tensors of those slots (a re-assigned parameter is seen) without walking the module tree (1 ms per forward)."""
        cache = self.__dict__.setdefault("_slot_cache", {})
        if tag not in cache:
            cache[tag] = [(m, n) for m in root.modules() for n in m._parameters if m._parameters[n] is not None]
        return [m._parameters[n] for m, n in cache[tag]]

    def _frozen_params(self):
        skip = set(id(p) for p in self._trainable_list())
        return [p for p in self._param_slots(self, "all") if id(p) not in skip]

    def _frozen_operands(self):
        """bf16 copies of the frozen weights; rebuilt only when a frozen tensor changed or moved."""
        key = tuple((p.data_ptr(), p._version) for p in self._frozen_params())
        if self._frozen_cache is not None and self._frozen_cache[0] == key:
            return self._frozen_cache[1]
        D, p = self.width, self.patch_size
        K = 3 * p * p
        Kp = (K + 63) // 64 * 64
        dev = self.conv1.weight.device
        wc = torch.zeros((D, Kp), dtype=F32, device=dev)
        wc[:, :K] = self.conv1.weight.detach().reshape(D, K).float()
        conv = torch.empty((D, Kp), dtype=BF16, device=dev)
        ops.cast_bf16(wc, conv)
        f = lambda t: t.detach().float().contiguous()
        out = dict(conv=conv, cls=f(self.class_embedding), pos=f(self.positional_embedding),
                   gpre=f(self.ln_pre.weight), bpre=f(self.ln_pre.bias),
                   blocks=[_Frozen(b) for b in self.transformer.resblocks])
        self._frozen_cache = (key, out)
        return out

    def set_inference_precision(self, precision: str):
        """'bf16' | 'fp8': operand type of the large GEMMs in no-grad forwards (training always runs bf16)."""
        if precision not in ('bf16', 'fp8'):
            raise ValueError("inference precision must be 'bf16' or 'fp8'")
        self.inference_precision = precision
        return self

    def _fp8_operands(self):
        """Per-block fp8 operands; rebuilt only when any block weight changed or moved."""
        # (FlatAdamW updates the adapters through raw pointers -- no tensor version changes -- and bumps `weights_epoch`)
        key = tuple((p.data_ptr(), p._version) for p in self._param_slots(self.transformer, "blocks")) + (self.weights_epoch,)
        if self._fp8_cache is None or self._fp8_cache[0] != key:
            self._fp8_cache = (key, [_Frozen8(b) for b in self.transformer.resblocks])
        return self._fp8_cache[1]

    def _stage_adapters(self, frozen, params):
        """Re-cast every adapter weight (fp32 master) into its persistent bf16 operand buffers with one
        ``aim_cast_multi`` launch.  The table of raw pointers is rebuilt only when a tensor moved."""
        srcs = []
        for i in range(self.layers):
            for j, a in enumerate(_ADAPTERS):
                k = 3 + (i * 3 + j) * 4
                srcs += [params[k], params[k + 2]]
                if a == "MLP_Adapter":
                    srcs.append(params[k + 1])
        ok = all(p.dtype == F32 and p.is_contiguous() for p in srcs)
        key = tuple(p.data_ptr() for p in srcs) + (id(frozen),)
        if ok and (self._cast_table is None or self._cast_table[0] != key):
            entries = []
            for i in range(self.layers):
                for j, a in enumerate(_ADAPTERS):
                    k = 3 + (i * 3 + j) * 4
                    entries += frozen["blocks"][i].cast_entries(a, params[k].detach(), params[k + 2].detach(),
                                                                params[k + 1].detach() if a == "MLP_Adapter" else None)
            self._cast_table = (key, ops.CastTable(entries, srcs[0].device))
        if ok:
            self._cast_table[1].run()
            return True
        for i in range(self.layers):           # generic path (non-fp32 / non-contiguous masters)
            for j, a in enumerate(_ADAPTERS):
                k = 3 + (i * 3 + j) * 4
                for src, dst, tr in frozen["blocks"][i].cast_entries(a, params[k].detach().float().contiguous(),
                                                                     params[k + 2].detach().float().contiguous()):
                    ops.cast_bf16(src, dst, transpose=tr)
        return False

    def _trainable_list(self):
        ps = [self.temporal_embedding, self.ln_post.weight, self.ln_post.bias]
        for blk in self.transformer.resblocks:
            for a in _ADAPTERS:
                m = getattr(blk, a)
                ps += [m.D_fc1.weight, m.D_fc1.bias, m.D_fc2.weight, m.D_fc2.bias]
        return ps

    @staticmethod
    def _drop_mask(N, drop_prob, scale, training, dev):
        """DropPath factor times adapter scale, per TOKEN index: timm's mask has shape (x.shape[0],1,1)
        and the reference's x is [N, BT, D] (vit_clip.py:112,275,286)."""
        if drop_prob > 0. and training:
            keep = 1.0 - drop_prob
            m = torch.empty(N, dtype=F32, device=dev).bernoulli_(keep)
            return m.div_(keep).mul_(scale) if keep > 0 else m.mul_(0.)
        return torch.full((N,), float(scale), dtype=F32, device=dev)

    def _drop_masks(self, N, training, dev):
        """Both DropPath factors of every block, ``[L, 2, N]`` (same distribution as ``_drop_mask`` per call; one
        uniform draw for the whole model instead of 2 L bernoulli launches)."""
        blocks = self.transformer.resblocks
        L = len(blocks)
        # the per-layer constants live on the device: building them on the host every step costs two pageable
        # host-to-device copies, each of which stalls the host until the stream has drained (no run-ahead across steps)
        key = (str(dev), tuple(float(b.drop_prob) for b in blocks), tuple(float(b.scale) for b in blocks))
        cached = getattr(self, "_drop_consts", None)
        if cached is None or cached[0] != key:
            rates = torch.tensor(key[1], dtype=F32)
            scale = torch.tensor(key[2], dtype=F32)
            keep = 1.0 - rates
            fac = torch.where(keep > 0, scale / keep.clamp_min(1e-12), torch.zeros_like(keep))
            cached = (key, float(rates.max()), scale.to(dev).view(L, 1, 1), keep.to(dev).view(L, 1, 1), fac.to(dev).view(L, 1, 1))
            self._drop_consts = cached
        _, max_rate, scale_d, keep_d, fac_d = cached
        if not training or max_rate <= 0.:
            return scale_d.expand(L, 2, N).contiguous()
        u = torch.rand((L, 2, N), dtype=F32, device=dev)
        return (u < keep_d).to(F32) * fac_d

    # ---- forward --------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("aim_amd.ViT_CLIP runs on MI355X only (HIP kernels); there is no CPU fallback")
        B, C, T, H, W = x.shape
        if T != self.num_frames:
            raise ValueError(f"expected {self.num_frames} frames, got {T}")   # reference: einops error at :443
        if C != 3 or H != self.input_resolution or W != self.input_resolution:
            raise ValueError(f"expected input [B,3,{T},{self.input_resolution},{self.input_resolution}], got {tuple(x.shape)}")
        if x.dtype == torch.float16:
            x = x.float()
        x = x.contiguous()
        # a fused GPUNormalize pre-hook arms the normalisation for THIS call (it runs before every forward it is registered
        # for): consumed here, so a removed hook or a caller that normalised the clip itself is not normalised twice
        self._norm_now = (self._norm_mean, self._norm_std) if x.dtype == torch.uint8 else (None, None)
        self._norm_mean = self._norm_std = None
        if x.dtype == torch.uint8 and self._norm_now[0] is None:
            raise TypeError("uint8 clips need a GPUNormalize module hook on the backbone (module_hooks.py:35-87)")
        if self.precision == 'fp32':
            from .fp32_path import _BackboneFn32, forward_f32
            if torch.is_grad_enabled() and any(p.requires_grad for p in self._trainable_list()):
                return _BackboneFn32.apply(self, x, *self._trainable_list()).unsqueeze(-1).unsqueeze(-1)
            with torch.no_grad():
                return forward_f32(self, x).unsqueeze(-1).unsqueeze(-1)
        y = _BackboneFn.apply(self, torch.is_grad_enabled(), x, *self._trainable_list())     # [B, D, T]
        return y.unsqueeze(-1).unsqueeze(-1)                          # BDTHW for I3D head (:456)
