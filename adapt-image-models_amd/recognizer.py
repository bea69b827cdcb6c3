"""Callers either side of the backbone, mirroring the reference's interfaces for this path only:

* ``Recognizer3D``  -- mmaction/models/recognizers/recognizer3d.py:8-118 + base.py:14-330
* ``I3DHead``       -- mmaction/models/heads/i3d_head.py:9-73 + heads/base.py:27-108
* ``CrossEntropyLoss`` -- mmaction/models/losses/cross_entropy_loss.py:9-80 (the hard-label branch, :78)
* ``top_k_accuracy``   -- mmaction/core/evaluation/accuracy.py:90-109
* ``GPUNormalize`` / ``register_module_hooks`` -- mmaction/utils/module_hooks.py:8-87 (fused into the
  patch-embedding kernel when the hooked module is this package's ``ViT_CLIP``).

These are thin host modules; the hot path is the backbone.  On GPU tensors the K400 training tail runs on
HIP kernels of libaim_hip.so: ``aim_head_fwd/bwd`` (avg-pool over frames + dropout + fc_cls)
and ``aim_ce_topk`` (hard-label cross-entropy + top-1/top-5 in one launch, no ``.cpu().numpy()`` sync per iteration,
heads/base.py:90).  ``_parse_losses`` reduces the log scalars in ONE all-reduce instead of four
(recognizers/base.py:237-242) and hands them back as lazily materialised floats (no host sync until a logger reads
them).  What the vit configs never use on this path -- soft labels, class weights, label smoothing, multi_class heads,
non-average pooling, feature extraction, gradcam, backward hooks -- is NOT restated here: those keywords raise
``NotImplementedError`` and belong to a real mmaction install (``register_into_mmaction``).  CPU tensors take a plain
``F.cross_entropy`` / ``nn.Linear`` (host-logic tests only); if the HIP library is missing, a GPU call raises
``LibraryNotBuilt`` -- it never falls back.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .registry import HEADS, LOSSES, RECOGNIZERS, Registry, build_backbone, build_head, build_loss

MODULE_HOOKS = Registry("module_hooks")


class LazyLogVars(OrderedDict):
    """``log_vars`` of ``_parse_losses``: an OrderedDict of Python floats that is filled in on FIRST READ.

    The reference calls ``.item()`` on every log variable in every iteration (recognizers/base.py:242): four host
    syncs per step.  Here the packed scalars start an asynchronous device-to-host copy into pinned memory when they
    are produced; whoever reads a value first (a logger hook every N iterations, a test) waits for that copy once."""

    def __init__(self, keys, packed: torch.Tensor):
        super().__init__((k, None) for k in keys)
        self._keys = list(keys)
        if packed.is_cuda:
            self._host = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
            self._host.copy_(packed, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._host, self._event = packed, None
        self._done = False

    def _materialise(self):
        if not self._done:
            self._done = True
            if self._event is not None:
                self._event.synchronize()
            for k, v in zip(self._keys, self._host.tolist()):
                OrderedDict.__setitem__(self, k, v)

    def __getitem__(self, k):
        self._materialise()
        return OrderedDict.__getitem__(self, k)

    def get(self, k, default=None):
        self._materialise()
        return OrderedDict.get(self, k, default)

    def items(self):
        self._materialise()
        return OrderedDict.items(self)

    def values(self):
        self._materialise()
        return OrderedDict.values(self)

    def __iter__(self):
        return OrderedDict.__iter__(self)

    def __repr__(self):
        self._materialise()
        return "LazyLogVars(%s)" % dict(OrderedDict.items(self))


def top_k_accuracy(scores, labels, topk=(1,)):
    """Reference semantics (accuracy.py:90-109): numpy argsort tie order."""
    res = []
    labels = np.array(labels)[:, np.newaxis]
    for k in topk:
        max_k_preds = np.argsort(scores, axis=1)[:, -k:][:, ::-1]
        match_array = np.logical_or.reduce(max_k_preds == labels, axis=1)
        res.append(match_array.sum() / match_array.shape[0])
    return res


def top_k_accuracy_device(cls_score: torch.Tensor, labels: torch.Tensor, topk=(1, 5)):
    """Same quantity without leaving the GPU.  A label is in the top k iff fewer than k classes score
    strictly higher, or tie and come later in numpy's stable ascending argsort (= larger index)."""
    s = cls_score.detach().float()
    tgt = s.gather(1, labels.view(-1, 1))
    idx = torch.arange(s.shape[1], device=s.device).view(1, -1)
    ahead = (s > tgt) | ((s == tgt) & (idx > labels.view(-1, 1)))
    rank = ahead.sum(1)
    return [(rank < min(k, s.shape[1])).float().mean() for k in topk]


class _HeadFn(torch.autograd.Function):
    """avg-pool over frames -> dropout factors -> fc_cls on ``aim_head_fwd`` / ``aim_head_bwd``."""

    @staticmethod
    def forward(ctx, feat, drop, W, b):
        from . import ops
        Wc = W.detach().float().contiguous()
        pooled, score = ops.head_fwd(feat.detach().float().contiguous(), drop, Wc, None if b is None else b.detach().float().contiguous())
        ctx.save_for_backward(pooled, Wc)
        ctx.drop, ctx.T, ctx.has_bias = drop, feat.shape[1], b is not None
        return score

    @staticmethod
    def backward(ctx, dscore):
        from . import ops
        pooled, Wc = ctx.saved_tensors
        dW, db, dfeat = ops.head_bwd(dscore.float().contiguous(), pooled, ctx.drop, Wc, ctx.T, need_dfeat=ctx.needs_input_grad[0])
        return dfeat, None, dW, (db if ctx.has_bias else None)


class _CETopkFn(torch.autograd.Function):
    """[mean CE, top-1, top-5] from ``aim_ce_topk``; only the CE slot carries a gradient."""

    @staticmethod
    def forward(ctx, score, label):
        from . import ops
        out3, dscore = ops.ce_topk(score.detach().float().contiguous(), label.reshape(-1).contiguous(), 5, need_grad=True)
        ctx.save_for_backward(dscore)
        return out3

    @staticmethod
    def backward(ctx, g3):
        (dscore,) = ctx.saved_tensors
        return dscore * g3[0], None


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """Hard-label cross entropy (cross_entropy_loss.py:78) times ``loss_weight`` (losses/base.py)."""

    def __init__(self, loss_weight=1.0, class_weight=None):
        super().__init__()
        if class_weight is not None:
            raise NotImplementedError("CrossEntropyLoss(class_weight=...) is outside the AIM ViT-CLIP path")
        self.loss_weight = loss_weight
        self.class_weight = None

    def forward(self, cls_score, label, **kwargs):
        if kwargs or cls_score.size() == label.size():
            raise NotImplementedError("soft labels / extra cross_entropy arguments are outside the AIM ViT-CLIP path")
        if cls_score.is_cuda and cls_score.dim() == 2 and label.dtype == torch.int64:
            return _CETopkFn.apply(cls_score, label)[0] * self.loss_weight      # aim_ce_topk
        return F.cross_entropy(cls_score, label) * self.loss_weight


@HEADS.register_module()
class I3DHead(nn.Module):
    """avg-pool over (T,H,W) -> dropout -> fc (i3d_head.py:53-73)."""

    def __init__(self, num_classes, in_channels, loss_cls=dict(type='CrossEntropyLoss'), spatial_type='avg',
                 dropout_ratio=0.5, init_std=0.01, multi_class=False, label_smooth_eps=0.0, **kwargs):
        super().__init__()
        if multi_class or label_smooth_eps != 0.0 or spatial_type != 'avg':
            raise NotImplementedError("I3DHead: multi_class / label_smooth_eps / spatial_type != 'avg' are outside the "
                                      "AIM ViT-CLIP path (every vit config uses the defaults)")
        self.num_classes, self.in_channels = num_classes, in_channels
        self.loss_cls = build_loss(loss_cls)
        self.multi_class, self.label_smooth_eps = False, 0.0
        self.spatial_type, self.dropout_ratio, self.init_std = spatial_type, dropout_ratio, init_std
        self.dropout = nn.Dropout(p=dropout_ratio) if dropout_ratio != 0 else None
        self.fc_cls = nn.Linear(in_channels, num_classes)

    def init_weights(self):
        nn.init.normal_(self.fc_cls.weight, 0, self.init_std)   # mmcv normal_init
        nn.init.constant_(self.fc_cls.bias, 0)

    def forward(self, x):
        # [B, C, T, H, W] -> [B, T*H*W, C]; the backbone's [B, D, T, 1, 1] output is a view of a frame-major [B, T, D]
        # buffer, so this is free.  Dropout is a caller-drawn factor table (same semantics as nn.Dropout on the pooled
        # [B, C] features: bernoulli(1 - p) / (1 - p)).
        feat = x.flatten(2).permute(0, 2, 1).contiguous()
        if x.is_cuda:
            drop = None
            if self.dropout is not None and self.training:
                keep = 1.0 - self.dropout_ratio
                drop = torch.empty((feat.shape[0], feat.shape[2]), dtype=torch.float32, device=x.device).bernoulli_(keep).div_(keep)
            return _HeadFn.apply(feat, drop, self.fc_cls.weight, self.fc_cls.bias)
        pooled = feat.mean(1)                                    # CPU tensors: host-logic tests
        if self.dropout is not None:
            pooled = self.dropout(pooled)
        return self.fc_cls(pooled)

    def loss(self, cls_score, labels, **kwargs):
        """heads/base.py:68-108 for hard labels: top-1 / top-5 accuracy + ``loss_cls``."""
        if kwargs:
            raise NotImplementedError("extra loss arguments are outside the AIM ViT-CLIP path")
        losses = dict()
        if labels.shape == torch.Size([]):
            labels = labels.unsqueeze(0)
        if cls_score.is_cuda and cls_score.dim() == 2 and labels.dtype == torch.int64:
            out3 = _CETopkFn.apply(cls_score, labels)        # CE + top-1/top-5 in one launch, nothing leaves the GPU
            losses['top1_acc'], losses['top5_acc'] = out3[1].detach(), out3[2].detach()
            losses['loss_cls'] = out3[0] * self.loss_cls.loss_weight
            return losses
        losses['top1_acc'], losses['top5_acc'] = top_k_accuracy_device(cls_score, labels, (1, 5))
        losses['loss_cls'] = self.loss_cls(cls_score, labels)
        return losses


@RECOGNIZERS.register_module()
class Recognizer3D(nn.Module):
    """3D recognizer framework for the AIM path (recognizer3d.py + recognizers/base.py)."""

    def __init__(self, backbone, cls_head=None, neck=None, train_cfg=None, test_cfg=None):
        super().__init__()
        if neck is not None:
            raise NotImplementedError("necks are outside the AIM ViT-CLIP path")
        self.backbone_from = 'mmaction2'
        self.backbone = build_backbone(backbone)
        self.cls_head = build_head(cls_head) if cls_head else None
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.aux_info = list(train_cfg['aux_info']) if train_cfg is not None and 'aux_info' in train_cfg else []
        self.max_testing_views = None
        if test_cfg is not None and 'max_testing_views' in test_cfg:
            self.max_testing_views = test_cfg['max_testing_views']
            assert isinstance(self.max_testing_views, int)
        if test_cfg and test_cfg.get('feature_extraction', False):
            raise NotImplementedError("test_cfg.feature_extraction is outside the AIM ViT-CLIP path")
        self.blending = None
        self.init_weights()
        self.fp16_enabled = False

    with_neck = False

    @property
    def with_cls_head(self):
        return self.cls_head is not None

    def init_weights(self):
        self.backbone.init_weights()
        if self.with_cls_head:
            self.cls_head.init_weights()

    def extract_feat(self, imgs):
        return self.backbone(imgs)

    def average_clip(self, cls_score, num_segs=1):
        """recognizers/base.py:160-194."""
        if 'average_clips' not in self.test_cfg.keys():
            raise KeyError('"average_clips" must defined in test_cfg\'s keys')
        average_clips = self.test_cfg['average_clips']
        if average_clips not in ['score', 'prob', None]:
            raise ValueError(f'{average_clips} is not supported. Currently supported ones are ["score", "prob", None]')
        if average_clips is None:
            return cls_score
        batch_size = cls_score.shape[0]
        cls_score = cls_score.view(batch_size // num_segs, num_segs, -1)
        if average_clips == 'prob':
            return F.softmax(cls_score, dim=2).mean(dim=1)
        return cls_score.mean(dim=1)

    def forward_train(self, imgs, labels, **kwargs):
        assert self.with_cls_head
        imgs = imgs.reshape((-1,) + imgs.shape[2:])
        x = self.extract_feat(imgs)
        cls_score = self.cls_head(x)
        gt_labels = labels.squeeze()
        return dict(self.cls_head.loss(cls_score, gt_labels, **kwargs))

    def _do_test(self, imgs):
        num_segs = imgs.shape[1]
        imgs = imgs.reshape((-1,) + imgs.shape[2:])
        if self.max_testing_views is not None:
            total_views = imgs.shape[0]
            assert num_segs == total_views, 'max_testing_views is only compatible with batch_size == 1'
            feats = [self.extract_feat(imgs[p:p + self.max_testing_views])
                     for p in range(0, total_views, self.max_testing_views)]
            feat = torch.cat(feats)
        else:
            feat = self.extract_feat(imgs)
        assert self.with_cls_head
        return self.average_clip(self.cls_head(feat), num_segs)

    def forward_test(self, imgs):
        return self._do_test(imgs).cpu().numpy()

    def forward_dummy(self, imgs, softmax=False):
        assert self.with_cls_head
        imgs = imgs.reshape((-1,) + imgs.shape[2:])
        outs = self.cls_head(self.extract_feat(imgs))
        if softmax:
            outs = F.softmax(outs, dim=-1)
        return (outs,)

    @staticmethod
    def _parse_losses(losses):
        """recognizers/base.py:211-244, with the per-variable all-reduces coalesced into one."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        packed = torch.stack([v.detach().float().reshape(()) for v in log_vars.values()])
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(packed.div_(dist.get_world_size()))
        return loss, LazyLogVars(list(log_vars.keys()), packed)

    def forward(self, imgs, label=None, return_loss=True, **kwargs):
        if kwargs.get('gradcam', False):
            raise NotImplementedError("gradcam is outside the AIM ViT-CLIP path")
        if return_loss:
            if label is None:
                raise ValueError('Label should not be None.')
            return self.forward_train(imgs, label, **kwargs)
        return self.forward_test(imgs, **kwargs)

    def train_step(self, data_batch, optimizer=None, **kwargs):
        imgs, label = data_batch['imgs'], data_batch['label']
        aux = {k: data_batch[k] for k in self.aux_info}
        losses = self(imgs, label, return_loss=True, **aux)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(next(iter(data_batch.values()))))

    val_step = train_step


@MODULE_HOOKS.register_module()
class GPUNormalize:
    """uint8 -> (x - mean) / std on the device (module_hooks.py:35-87).  On this package's ``ViT_CLIP``
    the normalisation is fused into the patch-embedding gather (``aim_patchify``) instead of
    materialising a float copy of the clip."""

    def __init__(self, input_format, mean, std):
        if input_format not in ['NCTHW', 'NCHW', 'NCHW_Flow', 'NPTCHW']:
            raise ValueError(f'The input format {input_format} is invalid.')
        self.input_format = input_format
        self.mean, self.std = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)
        shape = {'NCTHW': (1, -1, 1, 1, 1), 'NCHW': (1, -1, 1, 1), 'NCHW_Flow': (1, -1, 1, 1),
                 'NPTCHW': (1, 1, 1, -1, 1, 1)}[input_format]
        self._mean, self._std = self.mean.view(shape), self.std.view(shape)

    def hook_func(self):
        def normalize_hook(module, input):
            x = input[0]
            assert x.dtype == torch.uint8, (
                f'The previous augmentation should use uint8 data type to speed up computation, but get {x.dtype}')
            from .backbone import ViT_CLIP
            if isinstance(module, ViT_CLIP) and self.input_format == 'NCTHW':
                module._norm_mean = self.mean.to(x.device)
                module._norm_std = self.std.to(x.device)
                return input          # uint8 goes straight into the fused kernel
            with torch.no_grad():
                x = x.float().sub_(self._mean.to(x.device)).div_(self._std.to(x.device))
            return (x, *input[1:])
        return normalize_hook


def register_module_hooks(Module, module_hooks_list):
    """module_hooks.py:8-32."""
    handles = []
    for module_hook_cfg in module_hooks_list:
        cfg = dict(module_hook_cfg)
        hooked_module_name = cfg.pop('hooked_module', 'backbone')
        hook_pos = cfg.pop('hook_pos', 'forward_pre')
        module_hook = MODULE_HOOKS.build(cfg)
        hooked_module = getattr(Module, hooked_module_name)
        if hook_pos == 'forward_pre':
            handle = hooked_module.register_forward_pre_hook(module_hook.hook_func())
        elif hook_pos == 'forward':
            handle = hooked_module.register_forward_hook(module_hook.hook_func())
        else:
            raise ValueError(f'hook_pos must be `forward_pre` or `forward` here (backward hooks: use mmaction), but get {hook_pos}')
        handles.append(handle)
    return handles
