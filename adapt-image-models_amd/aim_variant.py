"""Stock-AIM backbone (``AIM``) on the HIP kernels: the published AIM model (README accuracy table), SURVEY section 8f-4.

Drop-in for ``mmaction/models/backbones/vitclip_aim.py:353-493`` (class ``AIM``) with ``wind_attn=False`` and
``num_tadapter=1`` -- the block at ``:195-211``:

    xt = T_Adapter(attention(ln_1('n (b t) d -> t (b n) d' x)))     temporal attention over the T frames of EVERY token
    x  = x + drop_path(xt)                                           (batch B*N; T_Adapter skip_connect=False)
    x  = x + S_Adapter(attention(ln_1(x)))                           S_Adapter skip_connect=True: y + fc2(gelu(fc1(y)))
    x  = x + mlp(ln_2(x)) + drop_path(scale * MLP_Adapter(ln_2(x)))  (same joint adaptation as vit_clip.py:285-286)

Same parameter names / shapes, same ``init_weights`` policy, same embedding and class-token readout as ``ViT_CLIP``
(``vitclip_aim.py:468-493`` restates ``vit_clip.py:433-458``), so it subclasses it; only the block's forward and its
hand-written backward differ.  Every GEMM here has M = B*T*N rows (persistent 256x256 kernel); the temporal attention is
``aim_tattn_fwd/bwd`` (csrc/tattn.hip), which reads the frame-major fused qkv buffer in place instead of rearranging the
activations twice as the reference does.  The window-attention branch (``wind_attn=True``, ``:213-285``) is a different
fork-only experiment and is not built.
"""
from typing import Dict, Optional

import torch

from . import ops
from .backbone import (BF16, F32, ViT_CLIP, _AdapterW, _Fork, _Frozen, _empty, _mlp_adapter_backward, _mlp_adapter_forward)
from .registry import BACKBONES


def aim_block_forward(x, fz: _Frozen, adp: Dict[str, _AdapterW], B, T, N, H, dp1, dms2, save: bool):
    """x: [B*T*N, D] f32 frame-major -> x3.  ``dp1``: the first DropPath factor per token (no adapter scale,
    vitclip_aim.py:205); ``dms2``: the second one times ``scale`` (:210)."""
    dev = x.device
    M, D = x.shape
    BT, r = B * T, fz.r
    tad, sad = adp["T_Adapter"], adp["S_Adapter"]
    # ---- temporal adaptation: ln_1 -> QKV -> attention over frames -> out_proj -> T_Adapter -> + drop_path
    xl = _empty((M, D), BF16, dev)
    mean1, rstd1 = _empty((M,), F32, dev), _empty((M,), F32, dev)
    ops.layernorm_fwd(x, fz.g1, fz.b1, M, D, D, y_bf16=xl, mean=mean1, rstd=rstd1)
    qkv_t = _empty((M, 3 * D), BF16, dev)
    ops.gemm(xl, fz.Wqkv, ops.EPI_BF16, qkv_t, bias=fz.bqkv)
    ot = _empty((M, D), BF16, dev)
    probs = _empty((B * N, H, T, T), F32, dev)
    ops.tattn_fwd(qkv_t, ot, probs, B, T, N, H)
    ta = _empty((M, D), BF16, dev)
    ops.gemm(ot, fz.Wo, ops.EPI_BF16, ta, bias=fz.bo)
    del ot, xl
    # the DropPath factor is folded into the stored activation (t_hs = dp1 * GELU(pre)), so the D_fc2 weight gradient is
    # a plain product and the bias rides along token-scaled as `vec`
    t_pre, t_hs = _empty((M, r), BF16, dev), _empty((M, r), BF16, dev)
    ops.gemm(ta, tad.W1, ops.EPI_ACT, t_hs, bias=tad.b1, out2=t_pre, act=ops.ACT_GELU, at=dp1, ntok=N)
    x1 = _empty((M, D), F32, dev)
    ops.gemm(t_hs, tad.W2, ops.EPI_F32, x1, resid=x, vec=tad.b2.reshape(1, -1), ldv=0, bt=dp1, ntok=N)
    # ---- spatial adaptation: ln_1 -> QKV -> attention over tokens -> out_proj -> S_Adapter (with skip)
    xl2 = _empty((M, D), BF16, dev)
    mean1b, rstd1b = _empty((M,), F32, dev), _empty((M,), F32, dev)
    ops.layernorm_fwd(x1, fz.g1, fz.b1, M, D, D, y_bf16=xl2, mean=mean1b, rstd=rstd1b)
    qkv_s = _empty((M, 3 * D), BF16, dev)
    ops.gemm(xl2, fz.Wqkv, ops.EPI_BF16, qkv_s, bias=fz.bqkv)
    del xl2
    ao = _empty((M, D), BF16, dev)
    lse = _empty((BT, H, N), F32, dev)
    ops.attn_fwd(qkv_s, ao, lse, BT, N, H)
    sa = _empty((M, D), BF16, dev)
    ops.gemm(ao, fz.Wo, ops.EPI_BF16, sa, bias=fz.bo)
    s_pre, s_h = _empty((M, r), BF16, dev), _empty((M, r), BF16, dev)
    ops.gemm(sa, sad.W1, ops.EPI_ACT, s_h, bias=sad.b1, out2=s_pre, act=ops.ACT_GELU)
    x2 = _empty((M, D), F32, dev)
    ops.gemm(s_h, sad.W2, ops.EPI_F32, x2, bias=sad.b2, resid=x1)      # x1 + D_fc2(GELU(D_fc1(sa)))
    ops.acc_bf16(x2, sa)                                                # + sa: the adapter's skip connection
    # ---- joint adaptation (shared with the vit_clip block)
    x3, xn, mean2, rstd2, hcat_pre, a_s = _mlp_adapter_forward(x2, fz, dms2, N, save)
    ctx = None
    if save:
        ctx = dict(x=x, mean1=mean1, rstd1=rstd1, qkv_t=qkv_t, probs=probs, ta=ta, t_pre=t_pre, t_hs=t_hs, x1=x1,
                   mean1b=mean1b, rstd1b=rstd1b, qkv_s=qkv_s, ao=ao, lse=lse, sa=sa, s_pre=s_pre, s_h=s_h, x2=x2,
                   mean2=mean2, rstd2=rstd2, xn=xn, hcat_pre=hcat_pre, a_s=a_s, dp1=dp1, dms2=dms2)
    return x3, ctx


def aim_block_backward(dyb, c, fz: _Frozen, adp: Dict[str, _AdapterW], grads, B, T, N, H, keep: Optional[list] = None):
    """dyb = d(loss)/d(x3) [M, D] bf16 -> d(loss)/d(x) bf16; the 12 adapter gradients are accumulated into ``grads``
    (their kernels run on the detached stream, joined at the end of the backward)."""
    dev = dyb.device
    M, D = dyb.shape
    BT, r = B * T, fz.r
    tad, sad = adp["T_Adapter"], adp["S_Adapter"]
    gS, gT = grads["S_Adapter"], grads["T_Adapter"]
    dx2b, later = _mlp_adapter_backward(dyb, c["x2"], c["mean2"], c["rstd2"], c["xn"], c["hcat_pre"], c["a_s"], c["dms2"],
                                        fz, grads["MLP_Adapter"], N)
    # ---- x2 = x1 + sa + (s_h W2^T + b2),  s_h = GELU(sa W1^T + b1),  sa = ao Wo^T + bo
    s_h, s_pre, sa = c["s_h"], c["s_pre"], c["sa"]
    later.append(lambda: ops.wgrad(dx2b, s_h, gS["D_fc2.weight"], gS["D_fc2.bias"]))
    dsh_pre = _empty((M, r), BF16, dev)
    ops.gemm(dx2b, sad.W2T, ops.EPI_DACT, dsh_pre, aux=s_pre, act=ops.ACT_GELU)
    later.append(lambda: ops.wgrad(dsh_pre, sa, gS["D_fc1.weight"], gS["D_fc1.bias"]))
    dsa = _empty((M, D), BF16, dev)
    ops.gemm(dsh_pre, sad.W1T, ops.EPI_BF16, dsa)
    ops.add_bf16(dsa, dx2b, dsa)                     # + the skip connection's share
    dao = _empty((M, D), BF16, dev)
    ops.gemm(dsa, fz.WoT, ops.EPI_BF16, dao)
    del dsa
    dqkv = _empty((M, 3 * D), BF16, dev)
    delta = _empty((BT, H, N), F32, dev)
    ops.attn_bwd(c["qkv_s"], c["ao"], dao, c["lse"], delta, dqkv, BT, N, H)
    del dao
    dxl2 = _empty((M, D), BF16, dev)
    ops.gemm(dqkv, fz.WqkvT, ops.EPI_BF16, dxl2)
    dx1b = _empty((M, D), BF16, dev)
    ops.layernorm_bwd(dxl2, c["x1"], fz.g1, c["mean1b"], c["rstd1b"], M, D, lddy=D, ldx=D, lddx=D, dres=dx2b, dx_bf16=dx1b)
    del dxl2
    # ---- x1 = x + t_hs W2^T + dp1[tok] b2,  t_hs = dp1[tok] GELU(ta W1^T + b1),  ta = attention_T(ln_1(x)) Wo^T + bo
    t_hs, t_pre, ta, dp1 = c["t_hs"], c["t_pre"], c["ta"], c["dp1"]
    later.append(lambda: ops.wgrad(dx1b, t_hs, gT["D_fc2.weight"], gT["D_fc2.bias"], at=dp1, ntok=N))
    dth_pre = _empty((M, r), BF16, dev)
    ops.gemm(dx1b, tad.W2T, ops.EPI_DACT, dth_pre, aux=t_pre, act=ops.ACT_GELU, at=dp1, ntok=N)
    later.append(lambda: ops.wgrad(dth_pre, ta, gT["D_fc1.weight"], gT["D_fc1.bias"]))
    dta = _empty((M, D), BF16, dev)
    ops.gemm(dth_pre, tad.W1T, ops.EPI_BF16, dta)
    dot = _empty((M, D), BF16, dev)
    ops.gemm(dta, fz.WoT, ops.EPI_BF16, dot)
    del dta
    ops.tattn_bwd(c["qkv_t"], c["probs"], dot, dqkv, B, T, N, H)      # (re-uses the spatial branch's d(qkv) buffer)
    del dot
    dxl = _empty((M, D), BF16, dev)
    ops.gemm(dqkv, fz.WqkvT, ops.EPI_BF16, dxl)
    del dqkv
    dxb = _empty((M, D), BF16, dev)
    ops.layernorm_bwd(dxl, c["x"], fz.g1, c["mean1"], c["rstd1"], M, D, lddy=D, ldx=D, lddx=D, dres=dx1b, dx_bf16=dxb)
    # the six weight gradients + bias column sums: off the gradient path, on the detached stream behind the main stream's
    # work so far
    fork = _Fork(dev, "aim-bwd")
    if fork.enabled:
        fork.started = True
        fork.sync_side_to_main()
    standalone = keep is None
    if standalone:
        keep = []
    fork.run_detached(later, keep)
    if standalone:
        _Fork.join_detached(dev)
    return dxb


@BACKBONES.register_module()
class AIM(ViT_CLIP):
    """Stock AIM (reference ``vitclip_aim.py:353-493``); constructor keywords of the reference class."""

    def __init__(self, input_resolution: int, num_frames: int, patch_size: int, width: int, layers: int, heads: int,
                 drop_path_rate, num_tadapter=1, adapter_scale=0.5, pretrained=None, prompt=True, wind_attn=False,
                 window_size=(32, 2, 2), not_shift=True):
        if wind_attn:
            raise NotImplementedError("AIM(wind_attn=True) (vitclip_aim.py:213-285, 3-D window attention) is not built; "
                                      "the stock AIM block is wind_attn=False")
        if num_tadapter != 1:
            raise NotImplementedError("AIM(num_tadapter=2) (T_Adapter_in, vitclip_aim.py:201-202) is not built")
        super().__init__(input_resolution, num_frames, patch_size, width, layers, heads, drop_path_rate,
                         adapter_scale=adapter_scale, pretrained=pretrained)
        self.variant = 'aim'
        self.num_tadapter, self.prompt, self.wind_attn = num_tadapter, prompt, wind_attn
