"""Frozen-backbone data parallelism: one process per GPU, adapter-only gradient all-reduce.

The reference wraps the recognizer in torch DDP (mmaction/apis/train.py:106-110), which buckets the
``requires_grad`` parameters (10 966 672 elements for ViT-B/16, T=8, 400 classes) into 25 MB NCCL
calls and all-reduces on every micro-step.  Here all trainable gradients live in ONE contiguous fp32
buffer (``param.grad`` are views into it), so a step is a single RCCL all-reduce over xGMI; with
gradient accumulation only the boundary micro-step communicates.  The frozen 86 M backbone weights
are never communicated after the initial broadcast.
"""
import os
from typing import Iterable, List

import torch
import torch.distributed as dist


def init_distributed(backend: str = None) -> tuple:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"    # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def broadcast_module(module: torch.nn.Module, src: int = 0):
    """Initial parameter/buffer broadcast from rank 0 (what the DDP constructor does, apis/train.py:106)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


class FlatGradReducer:
    """Owns one flat gradient buffer for ``params`` and mean-all-reduces it in a single collective."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("trainable parameters are kept in fp32 (bf16 operands are staged per step)")
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_grad(self):
        self.flat.zero_()
        off = 0
        for p in self.params:          # re-attach in case an optimizer replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * 4:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def all_reduce(self):
        """Mean over ranks, one call.  No-op for a single process."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat)
            self.flat.mul_(1.0 / dist.get_world_size())

    def broadcast_params(self, module: torch.nn.Module, src: int = 0):
        """Initial parameter broadcast from rank 0 (DDP constructor semantics)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src)


class FlatAdamW:
    """AdamW over flat buffers: parameters, gradients and both moments of all trainable tensors live in
    four contiguous fp32 buffers (``param.data`` / ``param.grad`` are views), grouped by (lr, weight_decay)
    so that a step is one ``aim_adamw_flat`` launch per group and one all-reduce for the whole model.

    ``groups`` is a list of dicts like torch's param groups: ``{"params": [...], "lr":, "weight_decay":}``.
    """

    def __init__(self, groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        from . import ops
        self._ops = ops
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        merged = {}
        for g in groups:
            key = (float(g.get("lr", lr)), float(g.get("weight_decay", weight_decay)))
            merged.setdefault(key, []).extend(p for p in g["params"] if p.requires_grad)
        self.param_groups = []
        params = []
        for (glr, gwd), ps in merged.items():
            self.param_groups.append(dict(params=ps, lr=glr, weight_decay=gwd, betas=betas, eps=eps))
            params += ps
        if not params:
            raise ValueError("no trainable parameters")
        dev = params[0].device
        pad = lambda n: (n + 3) // 4 * 4        # keep every tensor 16-byte aligned inside the flat buffers
        total = sum(pad(p.numel()) for p in params)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        off = 0
        self._views = []
        for grp in self.param_groups:
            grp["range"] = [off, off]
            for p in grp["params"]:
                if p.dtype != torch.float32:
                    raise TypeError("trainable parameters are kept in fp32")
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_p[off:off + n].view_as(p)
                p.grad = self.flat_g[off:off + n].view_as(p)
                self._views.append((p, off, n))
                off += pad(n)
            grp["range"][1] = off
        self.step_count = 0
        self.numel = total

    def zero_grad(self, set_to_none: bool = False):
        self.flat_g.zero_()
        for p, off, n in self._views:
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + off * 4:
                p.grad = self.flat_g[off:off + n].view_as(p)

    def all_reduce_grads(self):
        """Mean of the flat gradient over ranks: ONE RCCL all-reduce per step."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat_g)                       # SUM over xGMI (RCCL) ...
            self.flat_g.mul_(1.0 / dist.get_world_size())      # ... then the mean, one 44 MB elementwise pass

    def step(self):
        self.step_count += 1
        for grp in self.param_groups:
            a, b = grp["range"]
            if b > a:
                self._ops.adamw_flat(self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b], grp["lr"],
                                     grp["betas"][0], grp["betas"][1], grp["eps"], grp["weight_decay"], self.step_count)

    def state_dict(self):
        return dict(step=self.step_count, m=self.flat_m.clone(), v=self.flat_v.clone(),
                    groups=[dict(lr=g["lr"], weight_decay=g["weight_decay"], range=list(g["range"])) for g in self.param_groups])

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.flat_m.copy_(sd["m"]); self.flat_v.copy_(sd["v"])


def shard_indices(n: int, rank: int, world: int, seed: int = 0, epoch: int = 0, shuffle: bool = True):
    """DistributedSampler partition (mmaction/datasets/samplers/distributed_sampler.py:36-43): pad to a
    multiple of world, take indices[rank::world]; seed = epoch + seed."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(epoch + seed)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = (n + world - 1) // world * world
    idx = (idx * ((total + n - 1) // max(n, 1) + 1))[:total] if n else []
    return idx[rank:total:world]


def build_optimizer(model: torch.nn.Module, cfg: dict):
    """mmcv DefaultOptimizerConstructor semantics for the keys the vit configs use
    (configs/recognition/vit/vitclip_base_k400.py:96-102): AdamW + paramwise ``custom_keys`` decay_mult."""
    cfg = dict(cfg)
    typ = cfg.pop("type")
    paramwise = cfg.pop("paramwise_cfg", None) or {}
    custom = paramwise.get("custom_keys", {})
    base_wd = cfg.get("weight_decay", 0.0)
    base_lr = cfg["lr"]
    groups = []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g = {"params": [p]}
        for key in sorted(custom, key=len, reverse=True):
            if key in name:
                if "decay_mult" in custom[key]:
                    g["weight_decay"] = base_wd * custom[key]["decay_mult"]
                if "lr_mult" in custom[key]:
                    g["lr"] = base_lr * custom[key]["lr_mult"]
                break
        groups.append(g)
    if typ == "AdamW" and all(p.is_cuda for g in groups for p in g["params"]):
        opt = FlatAdamW(groups, lr=cfg["lr"], betas=tuple(cfg.get("betas", (0.9, 0.999))), eps=cfg.get("eps", 1e-8),
                        weight_decay=base_wd)
        for m in model.modules():            # the backbone may now add its gradients straight into the flat buffer
            if hasattr(m, "grad_in_place"):
                m.grad_in_place = True
        return opt
    opt_cls = getattr(torch.optim, typ)
    return opt_cls(groups, **cfg)
