"""CPU oracle for the AIM ViT-CLIP + Adapter hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the arithmetic in the
reference's ``mmaction/models/backbones/vit_clip.py`` and of the thin callers
either side of it.  It exists so that the HIP product path can be checked on a
GPU box where ``/root/reference`` does not exist.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product package (``adapt-image-models_amd/``) never does and
fails loudly when its HIP library is missing.

Parity pin: ``tests/golden/make_golden.py`` imports the real reference file
(in the build container only) and stores its outputs; ``tests/test_oracle_golden.py``
checks this restatement against those fixtures to <=1e-5 (fp32).

Two families of functions:

* ``ref_*``  – literal restatement, written the way the reference writes it
  (token-major ``[N, BT, D]`` tensors, duplicate LayerNorm, separate q/k/v
  matmuls ...).  Every function cites the reference lines it follows.
* ``emu_*``  – the same mathematics with the product's algebraic
  de-duplications and, when ``rnd`` is a bf16 rounding policy, a bf16 round-trip
  at exactly the points where the HIP path stores bf16.  With ``rnd=None`` it must
  agree with ``ref_*`` to fp32 round-off (tested); with ``rnd=BF16`` it is the
  tight comparator for the bf16 HIP path.

State dicts use the reference's parameter names (``vit_clip.py:335-350``):
``conv1.weight, class_embedding, positional_embedding, temporal_embedding,
ln_pre.*, transformer.resblocks.{i}.{attn.in_proj_weight, attn.in_proj_bias,
attn.out_proj.*, ln_1.*, ln_2.*, mlp.c_fc.*, mlp.c_proj.*,
{MLP,S,T}_Adapter.D_fc{1,2}.*}, ln_post.*``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


# --------------------------------------------------------------------------- #
# deterministic synthetic weights (shared by the golden generator and the tests)
# --------------------------------------------------------------------------- #
def backbone_param_shapes(input_resolution: int, num_frames: int, patch_size: int,
                          width: int, layers: int) -> Dict[str, Tuple[int, ...]]:
    """Parameter names/shapes of ``ViT_CLIP`` (``vit_clip.py:330-350, 85-118``)."""
    D = width
    r = int(D * 0.25)
    n_tok = (input_resolution // patch_size) ** 2 + 1
    s: Dict[str, Tuple[int, ...]] = {
        "conv1.weight": (D, 3, patch_size, patch_size),
        "class_embedding": (D,),
        "positional_embedding": (n_tok, D),
        "ln_pre.weight": (D,), "ln_pre.bias": (D,),
        "temporal_embedding": (1, num_frames, D),
        "ln_post.weight": (D,), "ln_post.bias": (D,),
    }
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        s[p + "attn.in_proj_weight"] = (3 * D, D)
        s[p + "attn.in_proj_bias"] = (3 * D,)
        s[p + "attn.out_proj.weight"] = (D, D)
        s[p + "attn.out_proj.bias"] = (D,)
        s[p + "ln_1.weight"] = (D,); s[p + "ln_1.bias"] = (D,)
        s[p + "mlp.c_fc.weight"] = (4 * D, D); s[p + "mlp.c_fc.bias"] = (4 * D,)
        s[p + "mlp.c_proj.weight"] = (D, 4 * D); s[p + "mlp.c_proj.bias"] = (D,)
        s[p + "ln_2.weight"] = (D,); s[p + "ln_2.bias"] = (D,)
        for a in ("MLP_Adapter", "S_Adapter", "T_Adapter"):
            s[p + a + ".D_fc1.weight"] = (r, D); s[p + a + ".D_fc1.bias"] = (r,)
            s[p + a + ".D_fc2.weight"] = (D, r); s[p + a + ".D_fc2.bias"] = (D,)
    return s


def _name_seed(name: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in (name + f"#{seed}").encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFF


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0,
                     live: bool = True) -> State:
    """Name-seeded random weights, independent of module construction order.

    ``live=True`` makes every term of the block contribute: ``D_fc2`` (zero in
    the reference's ``init_weights``, ``vit_clip.py:386-411``) and
    ``temporal_embedding`` (zeros, ``:344``) are randomised, LayerNorm gains
    are perturbed around 1 and biases are non-zero.  Scales are chosen so that
    activations stay O(1) through 12-24 layers.
    """
    out: State = {}
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed(_name_seed(name, seed))
        base = torch.randn(shape, generator=g, dtype=torch.float32)
        leaf = name.split(".")[-1]
        if name.startswith("ln_") or ".ln_" in name:
            t = 1.0 + 0.1 * base if leaf == "weight" else 0.05 * base
        elif leaf in ("bias", "in_proj_bias"):
            t = 0.02 * base
        elif name == "conv1.weight":
            t = base * (shape[1] * shape[2] * shape[3]) ** -0.5
        elif name in ("class_embedding", "positional_embedding"):
            t = base * 0.5
        elif name == "temporal_embedding":
            t = base * (0.3 if live else 0.0)
        elif "D_fc2.weight" in name:
            t = base * ((shape[1] ** -0.5) if live else 0.0)
        elif leaf in ("weight", "in_proj_weight"):
            t = base * shape[-1] ** -0.5
        else:
            t = 0.02 * base
        if (not live) and "D_fc2.bias" in name:
            t = torch.zeros(shape)
        out[name] = t.contiguous()
    return out


def trainable_names(state: State):
    """Freeze policy of ``ViT_CLIP.init_weights`` (``vit_clip.py:413-415``)."""
    return [n for n in state
            if ("temporal_embedding" in n or "ln_post" in n or "Adapter" in n or "cls_head" in n)]


# --------------------------------------------------------------------------- #
# literal restatement (ref_*)
# --------------------------------------------------------------------------- #
def ref_layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """``LayerNorm.forward`` (``vit_clip.py:71-77``): fp32 LN, eps 1e-5, cast back."""
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, 1e-5).to(x.dtype)


def ref_quick_gelu(x: Tensor) -> Tensor:
    """``QuickGELU`` (``vit_clip.py:80-82``)."""
    return x * torch.sigmoid(1.702 * x)


def ref_adapter(x: Tensor, st: State, pre: str) -> Tensor:
    """``Adapter.forward`` with ``skip_connect=False`` (``vit_clip.py:60-69, 105-108``)."""
    xs = F.linear(x, st[pre + ".D_fc1.weight"], st[pre + ".D_fc1.bias"])
    xs = F.gelu(xs)  # nn.GELU() exact erf (:52)
    return F.linear(xs, st[pre + ".D_fc2.weight"], st[pre + ".D_fc2.bias"])


def ref_attention(x: Tensor, y: Tensor, st: State, pre: str, heads: int,
                  need_weights: bool = False):
    """``attention`` / ``cross_attention`` (``vit_clip.py:128-197``).

    ``x`` supplies the queries ``[Tx, Nb, D]``; ``y`` the keys/values
    ``[Ty, Nb, D]`` (``y is x`` for ``attention``).  Returns ``out [Tx, Nb, D]``
    and, if asked, ``weights[Nb] = sum_{i,j} exp(sum_h aff)`` (``:149-151``).
    """
    W, bias = st[pre + "attn.in_proj_weight"], st[pre + "attn.in_proj_bias"]
    D = W.shape[1]
    dh = D // heads
    q = x @ W[:D].T + bias[:D]
    k = y @ W[D:-D].T + bias[D:-D]
    v = y @ W[-D:].T + bias[-D:]
    Tx, Ty, Nb = q.size(0), k.size(0), q.size(1)
    q = q.view(Tx, Nb, heads, dh).permute(1, 2, 0, 3)
    k = k.view(Ty, Nb, heads, dh).permute(1, 2, 0, 3)
    v = v.view(Ty, Nb, heads, dh).permute(1, 2, 0, 3)
    aff = q @ k.transpose(-2, -1) / (dh ** 0.5)
    weights = None
    if need_weights:
        with torch.no_grad():
            weights = torch.sum(torch.exp(torch.sum(aff, 1)).view(Nb, -1), -1)
    aff = aff.softmax(dim=-1)
    out = aff @ v
    out = out.permute(2, 0, 1, 3).flatten(2)
    out = F.linear(out, st[pre + "attn.out_proj.weight"], st[pre + "attn.out_proj.bias"])
    if need_weights:
        return out, weights
    return out


def ref_block(x: Tensor, st: State, i: int, heads: int, num_frames: int, scale: float,
              drop_mask: Optional[Tensor] = None, return_aux: bool = False):
    """``ResidualAttentionBlock.forward``, ``shift=False`` branch (``vit_clip.py:199-288``).

    ``x``: ``[N, BT, D]``.  ``drop_mask``: optional ``[N,1,1]`` DropPath factor
    (timm semantics: ``bernoulli(keep)/keep`` over ``x.shape[0]``, i.e. per token
    index; ``None`` = eval/identity), or a PAIR of such factors: the block calls its
    ``drop_path`` module twice (``:275`` and ``:286``) and every call draws a new mask.
    """
    pre = f"transformer.resblocks.{i}."
    n, bt, d = x.shape
    T = num_frames
    dm1, dm2 = drop_mask if isinstance(drop_mask, (tuple, list)) else (drop_mask, drop_mask)
    dp1 = (lambda t: t) if dm1 is None else (lambda t: t * dm1.reshape(-1, 1, 1))
    dp2 = (lambda t: t) if dm2 is None else (lambda t: t * dm2.reshape(-1, 1, 1))
    # temporal adaptation on the class tokens (:220-229)
    class_token = x[:1]                                           # 1, BT, D
    xt = class_token.reshape(1, bt // T, T, d).permute(2, 1, 0, 3).reshape(T, bt // T, d)
    ln1 = lambda t: ref_layer_norm(t, st[pre + "ln_1.weight"], st[pre + "ln_1.bias"])
    xt = ref_adapter(ref_attention(ln1(xt), ln1(xt), st, pre, heads), st, pre + "T_Adapter")
    xt = xt.reshape(T, bt // T, 1, d).permute(2, 1, 0, 3).reshape(1, bt, d)
    # spatial adaptation (:264-275)
    xl = ln1(x)
    ori_attn, ow = ref_attention(xl, xl, st, pre, heads, need_weights=True)
    crs_attn, cw = ref_attention(ln1(x), xt, st, pre, heads, need_weights=True)
    lamda = (cw / (cw + ow)).unsqueeze(0).unsqueeze(-1)
    x = x + (1 - lamda) * ori_attn + dp1(scale * ref_adapter(lamda * crs_attn, st, pre + "S_Adapter"))
    # joint adaptation (:285-286)
    xn = ref_layer_norm(x, st[pre + "ln_2.weight"], st[pre + "ln_2.bias"])
    h = F.linear(xn, st[pre + "mlp.c_fc.weight"], st[pre + "mlp.c_fc.bias"])
    h = F.linear(ref_quick_gelu(h), st[pre + "mlp.c_proj.weight"], st[pre + "mlp.c_proj.bias"])
    x = x + h + dp2(scale * ref_adapter(xn, st, pre + "MLP_Adapter"))
    if return_aux:
        return x, dict(ow=ow, cw=cw, lamda=lamda.reshape(-1), xt=xt.reshape(bt, d))
    return x


def ref_aim_block(x: Tensor, st: State, i: int, heads: int, num_frames: int, scale: float,
                  drop_mask=None) -> Tensor:
    """Stock-AIM ``ResidualAttentionBlock.forward``, ``wind_attn=False``, ``num_tadapter=1``
    (``mmaction/models/backbones/vitclip_aim.py:195-211``): temporal attention over the T frames of EVERY token
    position -> T_Adapter (no skip) -> DropPath; spatial attention -> S_Adapter WITH skip connection (``:124``,
    ``Adapter`` default ``skip_connect=True`` ``:76,93-96``), no scale, no DropPath; MLP + DropPath(scale * MLP_Adapter).
    ``x``: ``[N, BT, D]``; ``drop_mask``: None or the pair of ``[N]`` factors the block's two ``drop_path`` calls drew."""
    pre = f"transformer.resblocks.{i}."
    n, bt, d = x.shape
    T = num_frames
    dm1, dm2 = drop_mask if isinstance(drop_mask, (tuple, list)) else (drop_mask, drop_mask)
    dp1 = (lambda t: t) if dm1 is None else (lambda t: t * dm1.reshape(-1, 1, 1))
    dp2 = (lambda t: t) if dm2 is None else (lambda t: t * dm2.reshape(-1, 1, 1))
    ln1 = lambda t: ref_layer_norm(t, st[pre + "ln_1.weight"], st[pre + "ln_1.bias"])
    # temporal adaptation (:199-205): 'n (b t) d -> t (b n) d'
    xt = x.reshape(n, bt // T, T, d).permute(2, 1, 0, 3).reshape(T, (bt // T) * n, d)
    xt = ref_adapter(ref_attention(ln1(xt), ln1(xt), st, pre, heads), st, pre + "T_Adapter")
    xt = xt.reshape(T, bt // T, n, d).permute(2, 1, 0, 3).reshape(n, bt, d)      # 't (b n) d -> n (b t) d'
    x = x + dp1(xt)
    # spatial adaptation (:207): S_Adapter(y) = y + D_fc2(GELU(D_fc1(y)))
    sa = ref_attention(ln1(x), ln1(x), st, pre, heads)
    x = x + sa + ref_adapter(sa, st, pre + "S_Adapter")
    # joint adaptation (:209-210)
    xn = ref_layer_norm(x, st[pre + "ln_2.weight"], st[pre + "ln_2.bias"])
    h = F.linear(xn, st[pre + "mlp.c_fc.weight"], st[pre + "mlp.c_fc.bias"])
    h = F.linear(ref_quick_gelu(h), st[pre + "mlp.c_proj.weight"], st[pre + "mlp.c_proj.bias"])
    return x + h + dp2(scale * ref_adapter(xn, st, pre + "MLP_Adapter"))


def ref_aim_backbone(imgs: Tensor, st: State, heads: int, num_frames: int, scale: float = 0.5,
                     layers: Optional[int] = None, drop_masks=None) -> Tensor:
    """``AIM.forward`` (``vitclip_aim.py:468-493``): the embedding / ln_pre / ln_post / class-token readout are the
    same statements as ``vit_clip.py:433-458``; only the block differs."""
    B, _, T = imgs.shape[:3]
    if layers is None:
        layers = 1 + max(int(k.split(".")[2]) for k in st if k.startswith("transformer.resblocks."))
    x = ref_embed(imgs, st, num_frames)
    for i in range(layers):
        x = ref_aim_block(x, st, i, heads, num_frames, scale, drop_mask=None if drop_masks is None else drop_masks[i])
    x = x.permute(1, 0, 2)
    x = ref_layer_norm(x, st["ln_post.weight"], st["ln_post.bias"])
    x = x[:, 0]
    x = x.reshape(B, T, -1).permute(0, 2, 1)
    return x.unsqueeze(-1).unsqueeze(-1)


def ref_embed(imgs: Tensor, st: State, num_frames: int) -> Tensor:
    """``ViT_CLIP.forward`` up to ``ln_pre`` (``vit_clip.py:433-449``) -> ``[N, BT, D]``."""
    B, C, T, H, W = imgs.shape
    p = st["conv1.weight"].shape[-1]
    x = imgs.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    x = F.conv2d(x, st["conv1.weight"], None, stride=p)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = st["class_embedding"].to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype, device=x.device)
    x = torch.cat([cls, x], dim=1) + st["positional_embedding"].to(x.dtype)
    n = x.shape[1]
    x = x.reshape(B, T, n, -1).permute(0, 2, 1, 3).reshape(B * n, T, -1)
    x = x + st["temporal_embedding"]
    x = x.reshape(B, n, T, -1).permute(0, 2, 1, 3).reshape(B * T, n, -1)
    x = ref_layer_norm(x, st["ln_pre.weight"], st["ln_pre.bias"])
    return x.permute(1, 0, 2)


def ref_backbone(imgs: Tensor, st: State, heads: int, num_frames: int, scale: float = 0.5,
                 layers: Optional[int] = None, drop_masks=None) -> Tensor:
    """``ViT_CLIP.forward`` (``vit_clip.py:433-458``): ``[B,3,T,H,W] -> [B,D,T,1,1]``.
    ``drop_masks``: None (eval) or one ``(mask1, mask2)`` pair (or None) per layer (train mode)."""
    B, _, T = imgs.shape[:3]
    if layers is None:
        layers = 1 + max(int(k.split(".")[2]) for k in st if k.startswith("transformer.resblocks."))
    x = ref_embed(imgs, st, num_frames)
    for i in range(layers):
        x = ref_block(x, st, i, heads, num_frames, scale, drop_mask=None if drop_masks is None else drop_masks[i])
    x = x.permute(1, 0, 2)
    x = ref_layer_norm(x, st["ln_post.weight"], st["ln_post.bias"])
    x = x[:, 0]
    x = x.reshape(B, T, -1).permute(0, 2, 1)
    return x.unsqueeze(-1).unsqueeze(-1)


# callers either side of the path ------------------------------------------- #
def ref_i3d_head(feat: Tensor, fc_w: Tensor, fc_b: Tensor) -> Tensor:
    """``I3DHead.forward`` in eval / dropout off (``i3d_head.py:53-73``)."""
    x = F.adaptive_avg_pool3d(feat, (1, 1, 1))
    return F.linear(x.view(x.shape[0], -1), fc_w, fc_b)


def ref_cross_entropy(cls_score: Tensor, label: Tensor) -> Tensor:
    """``CrossEntropyLoss._forward`` hard-label path (``cross_entropy_loss.py:78``)."""
    return F.cross_entropy(cls_score, label)


def ref_top_k_accuracy(scores, labels, topk=(1,)):
    """``top_k_accuracy`` (``core/evaluation/accuracy.py:90-109``), numpy argsort order."""
    import numpy as np
    res = []
    labels = np.array(labels)[:, np.newaxis]
    for k in topk:
        max_k_preds = np.argsort(scores, axis=1)[:, -k:][:, ::-1]
        match_array = np.logical_or.reduce(max_k_preds == labels, axis=1)
        res.append(match_array.sum() / match_array.shape[0])
    return res


def ref_average_clip(cls_score: Tensor, num_segs: int, average_clips: Optional[str] = "prob") -> Tensor:
    """``BaseRecognizer.average_clip`` (``recognizers/base.py:160-194``)."""
    batch_size = cls_score.shape[0]
    cls_score = cls_score.view(batch_size // num_segs, num_segs, -1)
    if average_clips is None:
        return cls_score
    if average_clips == "prob":
        return F.softmax(cls_score, dim=2).mean(dim=1)
    if average_clips == "score":
        return cls_score.mean(dim=1)
    raise ValueError(average_clips)


def ref_gpu_normalize(imgs_u8: Tensor, mean, std) -> Tensor:
    """``GPUNormalize`` pre-hook (``utils/module_hooks.py:73-85``): uint8 -> (x-mean)/std."""
    shape = (1, 3, 1, 1, 1)
    m = torch.tensor(mean, dtype=torch.float32).view(shape)
    s = torch.tensor(std, dtype=torch.float32).view(shape)
    return (imgs_u8.float() - m) / s


# --------------------------------------------------------------------------- #
# product-dataflow restatement with optional bf16 rounding points (emu_*)
# --------------------------------------------------------------------------- #
class Rounding:
    """Where the HIP path stores bf16.  ``Rounding(None)`` = pure fp32."""

    def __init__(self, dtype: Optional[torch.dtype] = None):
        self.dtype = dtype

    def __call__(self, t: Tensor) -> Tensor:
        if self.dtype is None:
            return t
        # straight-through so autograd of the emulation stays well defined
        return t + (t.to(self.dtype).to(t.dtype) - t).detach()


FP32 = Rounding(None)
BF16 = Rounding(torch.bfloat16)


def q8(t: Tensor) -> Tensor:
    """Saturating round-trip through OCP fp8 e4m3 (what the inference kernels store as GEMM operands)."""
    return t.detach().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(t.dtype)


def q8_rows(w: Tensor):
    """Per-output-channel fp8 weight quantisation -> (dequantised-grid weight, scale) as ``ops.quantize_fp8_rows``."""
    scale = w.detach().abs().amax(dim=1).clamp_min(1e-12) / 448.0
    return q8(w.detach() / scale[:, None]), scale


def _lin(a: Tensor, w: Tensor, b: Optional[Tensor], rnd: Rounding) -> Tensor:
    """bf16-operand / fp32-accumulate GEMM: operands rounded, result left fp32."""
    return F.linear(rnd(a), rnd(w), b)


def emu_adapter(x: Tensor, st: State, pre: str, rnd: Rounding) -> Tensor:
    a = rnd(_lin(x, st[pre + ".D_fc1.weight"], st[pre + ".D_fc1.bias"], rnd))   # stored pre-activation
    h = rnd(F.gelu(a))
    return _lin(h, st[pre + ".D_fc2.weight"], st[pre + ".D_fc2.bias"], rnd)


def emu_block(x: Tensor, st: State, i: int, heads: int, T: int, scale: float,
              rnd: Rounding = FP32, drop_mask: Optional[Tensor] = None, return_aux: bool = False, f8: bool = False):
    """One block in the product's frame-major layout.

    ``x``: ``[BT, N, D]`` fp32 residual stream.  ``drop_mask``: ``[N]``, a pair of ``[N]``
    (one per ``drop_path`` call, ``vit_clip.py:275,286``) or None.
    Algebra: one ``ln_1``; one fused QKV projection whose class
    rows also feed the temporal attention; the cross-attention over a single
    key collapses to ``out_proj(W_v xt + b_v)`` broadcast over tokens; ``ow``/``cw``
    are full-width dot products ``exp(q_i . k_j / sqrt(dh))`` with a shared max
    shift (mathematically identical to ``vit_clip.py:151,186,272``).
    """
    pre = f"transformer.resblocks.{i}."
    BT, N, D = x.shape
    B = BT // T
    dh = D // heads
    W, bqkv = st[pre + "attn.in_proj_weight"], st[pre + "attn.in_proj_bias"]
    Wo, bo = st[pre + "attn.out_proj.weight"], st[pre + "attn.out_proj.bias"]
    xl_f = F.layer_norm(x, (D,), st[pre + "ln_1.weight"], st[pre + "ln_1.bias"], 1e-5)
    xl = rnd(xl_f)
    if f8:
        # inference path (BASELINE configs[4]): the four large GEMMs take fp8 e4m3 operands -- activations cast by the
        # producing kernel straight from fp32, weights quantised per output channel -- with fp32 accumulation; the
        # class-token chain keeps its own bf16 ln_1 + QKV projection on the B*T class rows
        def lin8(a_f32, w, b):
            w8, sc = q8_rows(w)
            return F.linear(q8(a_f32), w8) * sc + (0 if b is None else b)
        qkv = rnd(lin8(xl_f, W, bqkv))
        qkv_c = rnd(_lin(xl[:, 0], W, bqkv, rnd))
    else:
        qkv = rnd(_lin(xl, W, bqkv, rnd))                           # [BT,N,3D] stored bf16
        qkv_c = qkv[:, 0]
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    # --- temporal attention over the T class tokens of each clip ------------
    qc = qkv_c[:, :D].reshape(B, T, heads, dh).permute(0, 2, 1, 3)
    kc = qkv_c[:, D:2 * D].reshape(B, T, heads, dh).permute(0, 2, 1, 3)
    vc = qkv_c[:, 2 * D:].reshape(B, T, heads, dh).permute(0, 2, 1, 3)
    pt = ((qc @ kc.transpose(-1, -2)) / math.sqrt(dh)).softmax(-1)
    ot = rnd((pt @ vc).permute(0, 2, 1, 3).reshape(BT, D))
    ta = rnd(_lin(ot, Wo, bo, rnd))
    xt = rnd(emu_adapter(ta, st, pre + "T_Adapter", rnd))           # [BT, D]
    # --- cross term: single key/value per frame ------------------------------
    kx = rnd(_lin(xt, W[D:2 * D], bqkv[D:2 * D], rnd))
    vx = rnd(_lin(xt, W[2 * D:], bqkv[2 * D:], rnd))
    crs = _lin(vx, Wo, bo, rnd)                                     # [BT, D] fp32
    # --- lamda (no grad, :150,185,272) ----------------------------------------
    with torch.no_grad():
        s_full = (q.detach() @ k.detach().transpose(-1, -2)) / math.sqrt(dh)   # [BT,N,N]
        s_crs = (q.detach() @ kx.detach().unsqueeze(-1)).squeeze(-1) / math.sqrt(dh)  # [BT,N]
        m = torch.maximum(s_full.amax(dim=(1, 2)), s_crs.amax(dim=1))
        ow = torch.exp(s_full - m[:, None, None]).sum(dim=(1, 2))
        cw = torch.exp(s_crs - m[:, None]).sum(dim=1)
        lam = cw / (cw + ow)                                        # [BT]
    # --- spatial attention ------------------------------------------------------
    qh = q.reshape(BT, N, heads, dh).permute(0, 2, 1, 3)
    kh = k.reshape(BT, N, heads, dh).permute(0, 2, 1, 3)
    vh = v.reshape(BT, N, heads, dh).permute(0, 2, 1, 3)
    sc = (qh @ kh.transpose(-1, -2)) / math.sqrt(dh)
    pu = torch.exp(sc - sc.amax(dim=-1, keepdim=True))           # un-normalised, max 1 (what the kernel rounds)
    ao_f = ((rnd(pu) @ vh) / pu.sum(dim=-1, keepdim=True)).permute(0, 2, 1, 3).reshape(BT, N, D)
    if f8:
        proj = lin8(ao_f, Wo, bo)
    else:
        ao = rnd(ao_f)
        proj = _lin(ao, Wo, bo, rnd)                                # fp32 accumulators
    sv = emu_adapter(rnd(lam[:, None] * crs), st, pre + "S_Adapter", rnd)   # [BT, D]
    d1, d2 = drop_mask if isinstance(drop_mask, (tuple, list)) else (drop_mask, drop_mask)
    dm = 1.0 if d1 is None else d1.reshape(1, N, 1)
    dm2 = 1.0 if d2 is None else d2.reshape(1, N, 1)
    x1 = x + (1 - lam)[:, None, None] * proj + dm * scale * sv[:, None, :]
    if f8:
        x1 = rnd(x1)          # the fp8 inference path keeps its residual stream in bf16 (AIM_EPI_RES16: one rounding per update)
    # --- MLP + MLP_Adapter --------------------------------------------------------
    xn_f = F.layer_norm(x1, (D,), st[pre + "ln_2.weight"], st[pre + "ln_2.bias"], 1e-5)
    if f8:
        # concatenated operands: [W_fc ; D_fc1] along N, [W_proj | D_fc2] along K (one scale per output channel of each)
        ap = pre + "MLP_Adapter."
        h = ref_quick_gelu(lin8(xn_f, st[pre + "mlp.c_fc.weight"], st[pre + "mlp.c_fc.bias"]))
        a = dm2 * scale * F.gelu(lin8(xn_f, st[ap + "D_fc1.weight"], st[ap + "D_fc1.bias"]))
        wcat2 = torch.cat([st[pre + "mlp.c_proj.weight"], st[ap + "D_fc2.weight"]], dim=1)
        out2 = lin8(torch.cat([h, a], dim=-1), wcat2, None)
        x2 = rnd(x1 + out2 + st[pre + "mlp.c_proj.bias"] + dm2 * scale * st[ap + "D_fc2.bias"])
        if return_aux:
            return x2, dict(lamda=lam, xt=xt, ow_shifted=ow, cw_shifted=cw, shift=m)
        return x2
    xn = rnd(xn_f)
    hpre = rnd(_lin(xn, st[pre + "mlp.c_fc.weight"], st[pre + "mlp.c_fc.bias"], rnd))
    h = rnd(ref_quick_gelu(hpre))
    mlp = _lin(h, st[pre + "mlp.c_proj.weight"], st[pre + "mlp.c_proj.bias"], rnd)
    ad = emu_adapter(xn, st, pre + "MLP_Adapter", rnd)
    x2 = x1 + mlp + dm2 * scale * ad
    if return_aux:
        # ow/cw are reported un-shifted only through their ratio; lam is the contract
        return x2, dict(lamda=lam, xt=xt, ow_shifted=ow, cw_shifted=cw, shift=m)
    return x2


def emu_aim_block(x: Tensor, st: State, i: int, heads: int, T: int, scale: float, rnd: Rounding = FP32,
                  drop_mask=None) -> Tensor:
    """Stock-AIM block in the product's frame-major dataflow ``[BT, N, D]`` with the product's bf16 rounding points."""
    pre = f"transformer.resblocks.{i}."
    BT, N, D = x.shape
    B = BT // T
    dh = D // heads
    W, bqkv = st[pre + "attn.in_proj_weight"], st[pre + "attn.in_proj_bias"]
    Wo, bo = st[pre + "attn.out_proj.weight"], st[pre + "attn.out_proj.bias"]
    d1, d2 = drop_mask if isinstance(drop_mask, (tuple, list)) else (drop_mask, drop_mask)
    dm1 = 1.0 if d1 is None else d1.reshape(1, N, 1)
    dm2 = 1.0 if d2 is None else d2.reshape(1, N, 1)
    ln1 = lambda t: rnd(F.layer_norm(t, (D,), st[pre + "ln_1.weight"], st[pre + "ln_1.bias"], 1e-5))

    def attn(q, k, v):          # [..., S, heads*dh] over the second-to-last axis
        sh = q.shape[:-1]
        qh, kh, vh = (t.reshape(*sh, heads, dh).transpose(-2, -3) for t in (q, k, v))
        p = ((qh @ kh.transpose(-1, -2)) / math.sqrt(dh)).softmax(-1)
        return (p @ vh).transpose(-2, -3).reshape(*sh, heads * dh)

    # temporal: sequence = frames, batch = (clip, token)
    qkv = rnd(_lin(ln1(x), W, bqkv, rnd)).reshape(B, T, N, 3 * D).permute(0, 2, 1, 3)      # [B, N, T, 3D]
    ot = rnd(attn(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]))                     # [B, N, T, D]
    ot = ot.permute(0, 2, 1, 3).reshape(BT, N, D)
    ta = rnd(_lin(ot, Wo, bo, rnd))
    tp = pre + "T_Adapter"
    t_pre = rnd(_lin(ta, st[tp + ".D_fc1.weight"], st[tp + ".D_fc1.bias"], rnd))
    t_hs = rnd(dm1 * F.gelu(t_pre))                        # DropPath factor folded into the stored activation
    x1 = x + _lin(t_hs, st[tp + ".D_fc2.weight"], None, rnd) + dm1 * st[tp + ".D_fc2.bias"]
    # spatial, S_Adapter with skip
    qkv2 = rnd(_lin(ln1(x1), W, bqkv, rnd))
    q, k, v = qkv2[..., :D], qkv2[..., D:2 * D], qkv2[..., 2 * D:]
    qh = q.reshape(BT, N, heads, dh).permute(0, 2, 1, 3)
    kh = k.reshape(BT, N, heads, dh).permute(0, 2, 1, 3)
    vh = v.reshape(BT, N, heads, dh).permute(0, 2, 1, 3)
    sc = (qh @ kh.transpose(-1, -2)) / math.sqrt(dh)
    pu = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
    ao = rnd(((rnd(pu) @ vh) / pu.sum(dim=-1, keepdim=True)).permute(0, 2, 1, 3).reshape(BT, N, D))
    sa = rnd(_lin(ao, Wo, bo, rnd))
    x2 = x1 + sa + emu_adapter(sa, st, pre + "S_Adapter", rnd)
    # MLP + MLP_Adapter (as the vit_clip block)
    xn = rnd(F.layer_norm(x2, (D,), st[pre + "ln_2.weight"], st[pre + "ln_2.bias"], 1e-5))
    hpre = rnd(_lin(xn, st[pre + "mlp.c_fc.weight"], st[pre + "mlp.c_fc.bias"], rnd))
    h = rnd(ref_quick_gelu(hpre))
    mlp = _lin(h, st[pre + "mlp.c_proj.weight"], st[pre + "mlp.c_proj.bias"], rnd)
    mp = pre + "MLP_Adapter"
    a_pre = rnd(_lin(xn, st[mp + ".D_fc1.weight"], st[mp + ".D_fc1.bias"], rnd))
    a_s = rnd(dm2 * scale * F.gelu(a_pre))
    return x2 + mlp + _lin(a_s, st[mp + ".D_fc2.weight"], None, rnd) + dm2 * scale * st[mp + ".D_fc2.bias"]


def emu_aim_backbone(imgs: Tensor, st: State, heads: int, scale: float = 0.5, rnd: Rounding = FP32,
                     layers: Optional[int] = None, drop_masks=None) -> Tensor:
    B, _, T = imgs.shape[:3]
    if layers is None:
        layers = 1 + max(int(k.split(".")[2]) for k in st if k.startswith("transformer.resblocks."))
    x = emu_embed(imgs, st, rnd)
    for i in range(layers):
        x = emu_aim_block(x, st, i, heads, T, scale, rnd, drop_mask=None if drop_masks is None else drop_masks[i])
    c = F.layer_norm(x[:, 0], (x.shape[-1],), st["ln_post.weight"], st["ln_post.bias"], 1e-5)
    return c.reshape(B, T, -1).permute(0, 2, 1).unsqueeze(-1).unsqueeze(-1)


def emu_embed(imgs: Tensor, st: State, rnd: Rounding = FP32) -> Tensor:
    """Patch embed + cls/pos/temporal + ``ln_pre`` -> ``[BT, N, D]`` fp32."""
    B, C, T, H, W = imgs.shape
    D, _, p, _ = st["conv1.weight"].shape
    g = H // p
    patches = imgs.permute(0, 2, 1, 3, 4).reshape(B * T, C, g, p, g, p)
    patches = patches.permute(0, 2, 4, 1, 3, 5).reshape(B * T, g * g, C * p * p)
    tok = rnd(_lin(patches, st["conv1.weight"].reshape(D, -1), None, rnd))   # [BT, g*g, D]
    cls = st["class_embedding"].expand(B * T, 1, D)
    x = torch.cat([cls, tok], dim=1) + st["positional_embedding"]
    x = (x.reshape(B, T, -1, D) + st["temporal_embedding"].reshape(1, T, 1, D)).reshape(B * T, -1, D)
    return F.layer_norm(x, (D,), st["ln_pre.weight"], st["ln_pre.bias"], 1e-5)


def emu_backbone(imgs: Tensor, st: State, heads: int, scale: float = 0.5,
                 rnd: Rounding = FP32, layers: Optional[int] = None, drop_masks=None, f8: bool = False) -> Tensor:
    """Whole backbone in the product's dataflow: ``[B,3,T,H,W] -> [B,D,T,1,1]``."""
    B, _, T = imgs.shape[:3]
    if layers is None:
        layers = 1 + max(int(k.split(".")[2]) for k in st if k.startswith("transformer.resblocks."))
    x = emu_embed(imgs, st, rnd)
    if f8:
        x = rnd(x)
    for i in range(layers):
        x = emu_block(x, st, i, heads, T, scale, rnd, drop_mask=None if drop_masks is None else drop_masks[i], f8=f8)
    c = F.layer_norm(x[:, 0], (x.shape[-1],), st["ln_post.weight"], st["ln_post.bias"], 1e-5)
    return c.reshape(B, T, -1).permute(0, 2, 1).unsqueeze(-1).unsqueeze(-1)
