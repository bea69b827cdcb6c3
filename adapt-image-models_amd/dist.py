"""Frozen-backbone data parallelism: one process per GPU, adapter-only gradient all-reduce.

The reference wraps the recognizer in torch DDP (mmaction/apis/train.py:106-110), which buckets the
``requires_grad`` parameters (10 966 672 elements for ViT-B/16, T=8, 400 classes) into 25 MB NCCL
calls that fire during backward, on every micro-step.  Here all trainable gradients live in ONE contiguous fp32
buffer (``param.grad`` are views into it) and a step reduces it in at most three RCCL calls over xGMI: the slices of the
upper two thirds of the layers start on a communication stream as soon as the backward has queued their last
producers (``ViT_CLIP.grad_ready_hook``), overlapped with the backward of the lower layers; the rest follows when
backward returns.  The collective is a SUM; the 1/world of DDP's mean is folded into the AdamW kernel
(``aim_adamw_flat(grad_scale=)``), so there is no separate pass over the buffer.  With gradient accumulation only the
boundary micro-step communicates (``FlatAdamW.no_sync()``).  The frozen 86 M backbone weights are never communicated
after the initial broadcast.
"""
import contextlib
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
        # overlapped reduction state (see attach_backbone)
        self._buckets = []           # [(first_layer, a, b)]: flat range [a, b) is complete once `first_layer`'s backward is queued
        self._works = []
        self._reduced = []           # ranges already handed to the collective in this step
        self._sync = True
        self._summed = False         # flat_g holds the SUM over ranks (this step's collective has run)
        self._comm = None
        self.early_launches = 0      # diagnostics: collectives started from inside backward
        # early buckets start from inside backward (the caller promises ONE backward per synchronised step; more than one
        # raises, see _on_layer_queued).  False / AIM_DP_OVERLAP=0: one reduction inside step() after any number of backwards
        self.overlap = os.environ.get("AIM_DP_OVERLAP", "1") != "0"
        self._top_layer = -1
        self._backbones = []         # attached backbones: told when their trainable weights changed (fp8 operand caches)
        self.force_collectives = False

    # ---- overlapped all-reduce ---------------------------------------------------------------------------------
    @staticmethod
    def _world():
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def _active(self):
        """Collectives are issued when there is more than one rank -- or, for tests that must drive the real RCCL call
        pattern on a one-GPU box, when ``force_collectives`` is set on an initialised single-rank group."""
        return self._world() > 1 or (self.force_collectives and dist.is_available() and dist.is_initialized())

    def attach_backbone(self, backbone, prefix_of=None, n_buckets: int = 3):
        """Plan early buckets over the flat buffer from the backbone's layer order and install its
        ``grad_ready_hook``.  Layers [L*(k-1)/n, L*k/n) form bucket k; bucket k (k >= 1) fires when the backward of its
        lowest layer has been queued, bucket 0 (lowest layers + everything else) after backward.  A bucket is used only
        if its parameters occupy ONE contiguous range of the flat buffer holding nothing else."""
        blocks = list(backbone.transformer.resblocks)
        L = len(blocks)
        pos = {id(p): (off, n) for p, off, n in self._views}
        pad = lambda n: (n + 3) // 4 * 4
        self._buckets = []
        for k in range(n_buckets - 1, 0, -1):
            lo, hi = L * k // n_buckets, L * (k + 1) // n_buckets
            ps = [p for blk in blocks[lo:hi] for p in blk.parameters() if p.requires_grad]
            if not ps or any(id(p) not in pos for p in ps):
                continue
            a = min(pos[id(p)][0] for p in ps)
            b = max(pos[id(p)][0] + pad(pos[id(p)][1]) for p in ps)
            if b - a == sum(pad(pos[id(p)][1]) for p in ps):
                self._buckets.append((lo, a, b))
        backbone.grad_ready_hook = self._on_layer_queued
        self._top_layer = L - 1
        self._backbones.append(backbone)
        return self._buckets

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: micro-steps inside do not communicate (DDP.no_sync semantics)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _on_layer_queued(self, layer: int, in_place: bool, streams):
        if not self._sync or not self._active() or not in_place or not self.overlap:
            return
        if layer == self._top_layer and self._reduced:
            # DDP reduces on every backward; here a step's gradients are reduced ONCE (the mean of the sum = the sum of
            # the means), so a backward that arrives after the reduction has started would add local gradients to
            # already-summed buckets.  Never silently: the reference's accumulation contract (update_interval > 1,
            # mmaction/utils/optimizer.py:22-33) is served by no_sync() / aim_amd.DistOptimizerHook or overlap = False.
            raise RuntimeError("FlatAdamW: backward() after this step's gradient reduction has started; wrap the "
                               "non-boundary micro-steps in opt.no_sync() (aim_amd.DistOptimizerHook does) or set "
                               "opt.overlap = False (one reduction inside step(), any number of backwards)")
        for lo, a, b in self._buckets:
            if lo == layer and (a, b) not in self._reduced:
                self._launch(a, b, streams)

    def _launch(self, a: int, b: int, streams):
        dev = self.flat_g.device
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=dev)
        for st in streams:                       # behind everything queued so far on the producing streams
            ev = torch.cuda.Event()
            ev.record(st)
            self._comm.wait_event(ev)
        with torch.cuda.stream(self._comm):
            self._works.append(dist.all_reduce(self.flat_g[a:b], async_op=True))
        self._reduced.append((a, b))
        self.early_launches += 1

    def zero_grad(self, set_to_none: bool = False):
        for w in self._works:                           # (a reduction nobody waited for must not land in the zeroed buffer)
            w.wait()
        self._works, self._reduced, self._summed = [], [], False
        self.flat_g.zero_()
        for p, off, n in self._views:
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + off * 4:
                p.grad = self.flat_g[off:off + n].view_as(p)

    def all_reduce_grads(self):
        """Finish the step's gradient reduction (SUM over ranks; the mean's 1/world is applied inside ``step``): reduce
        whatever the early buckets did not cover -- everything, when no backbone is attached -- and make the current
        stream wait for all of it.  Idempotent within a step: a second call finds nothing left.  After it ``flat_g`` /
        ``param.grad`` hold the SUM over ranks (DDP leaves the mean there): ``grad_scale`` is the factor that turns
        them into DDP's mean, and ``clip_grad_norm_`` accounts for it."""
        if not self._active() or not self._sync:
            return
        todo, cur = [], 0
        for a, b in sorted(self._reduced):
            if a > cur:
                todo.append((cur, a))
            cur = max(cur, b)
        if cur < self.numel:
            todo.append((cur, self.numel))
        for a, b in todo:
            self._works.append(dist.all_reduce(self.flat_g[a:b], async_op=True))
            self._reduced.append((a, b))
        for w in self._works:
            w.wait()                                    # stream-ordered on GPU backends: no host sync
        self._works = []
        self._summed = True                             # flat_g now holds the SUM over ranks

    @property
    def grad_scale(self) -> float:
        """What turns ``param.grad`` into the data-parallel MEAN gradient right now: 1/world after this step's
        all-reduce ran (the collective is a SUM), 1 before it or inside ``no_sync()`` (local gradients)."""
        return 1.0 / self._world() if self._summed else 1.0

    def clip_grad_norm_(self, max_norm: float, norm_type: float = 2.0):
        """mmcv ``OptimizerHook(grad_clip=dict(max_norm=...))`` on the flat buffer (the reference's sgd schedules use
        ``max_norm=40``): the norm is taken of the MEAN gradient (``grad_scale`` applied), like
        ``torch.nn.utils.clip_grad_norm_`` sees it under DDP.  Finishes a pending reduction first; returns the norm."""
        self.all_reduce_grads()
        total = torch.linalg.vector_norm(self.flat_g, norm_type) * self.grad_scale
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        self.flat_g.mul_(coef)
        return total

    def step(self):
        """One AdamW step.  The reference's hook contract is ``loss.backward(); optimizer.step()``
        (mmaction/utils/optimizer.py:22-33: DDP has finished its buckets when backward returns): a step therefore first
        launches whatever part of the gradient reduction is still outstanding and waits for all of it (stream-ordered),
        so a caller that never calls ``all_reduce_grads()`` itself still steps on fully reduced gradients."""
        self.all_reduce_grads()
        self.step_count += 1
        gs = self.grad_scale                            # SUM all-reduce -> mean, inside the optimizer kernel
        for grp in self.param_groups:
            a, b = grp["range"]
            if b > a:
                self._ops.adamw_flat(self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b], grp["lr"],
                                     grp["betas"][0], grp["betas"][1], grp["eps"], grp["weight_decay"], self.step_count,
                                     grad_scale=gs)
        for bb in self._backbones:                      # the update went through raw pointers: no tensor version changed
            bb.weights_epoch = getattr(bb, "weights_epoch", 0) + 1

    def state_dict(self):
        """torch.optim-shaped: ``state`` = per-parameter {step, exp_avg, exp_avg_sq} keyed by the parameter's index in
        ``param_groups`` order, ``param_groups`` = hyper-parameters + index lists -- so a checkpoint written here loads
        into ``torch.optim.AdamW`` over the same parameter order and vice versa (mmcv's checkpoint hook stores
        ``optimizer.state_dict()`` as is: mmcv_custom/runner/checkpoint.py:39-80)."""
        pos = {id(p): (off, n) for p, off, n in self._views}
        state, groups, idx = {}, [], 0
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                off, n = pos[id(p)]
                if self.step_count > 0:
                    state[idx] = dict(step=torch.tensor(float(self.step_count)),
                                      exp_avg=self.flat_m[off:off + n].view_as(p).clone(),
                                      exp_avg_sq=self.flat_v[off:off + n].view_as(p).clone())
                ids.append(idx)
                idx += 1
            # the full key set of torch.optim.AdamW's groups: a loader that takes the saved group as is (torch does) must
            # find decoupled_weight_decay=True, or it would run Adam with L2 regularisation instead
            groups.append(dict(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"],
                               amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                               fused=None, decoupled_weight_decay=True, params=ids))
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        pos = {id(p): (off, n) for p, off, n in self._views}
        flat = [p for g in self.param_groups for p in g["params"]]
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            if len(sg["params"]) != len(g["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
            for k in ("lr", "weight_decay", "eps"):
                if k in sg:
                    g[k] = float(sg[k])
            if "betas" in sg:
                g["betas"] = tuple(sg["betas"])
        steps = set()
        for idx, st in sd["state"].items():
            p = flat[int(idx)]
            off, n = pos[id(p)]
            self.flat_m[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.flat_v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FlatAdamW keeps one step count for all parameters")
        self.step_count = steps.pop() if steps else 0


class DistOptimizerHook:
    """The reference's optimizer hook (mmaction/utils/optimizer.py:9-33) for ``FlatAdamW``: ``loss /= update_interval``,
    backward on every iteration, clip + step + zero_grad on every ``update_interval``-th.  Same constructor keywords
    (``coalesce`` / ``bucket_size_mb`` are accepted and unused there too; ``use_fp16`` selected apex loss scaling, which
    the bf16 policy does not need).  Unlike the reference -- whose DDP all-reduces on every micro-step -- only the
    boundary micro-step communicates: the others run inside ``no_sync()``."""

    def __init__(self, update_interval=1, grad_clip=None, coalesce=True, bucket_size_mb=-1, use_fp16=False):
        self.grad_clip = grad_clip
        self.coalesce = coalesce
        self.bucket_size_mb = bucket_size_mb
        self.update_interval = update_interval
        self.use_fp16 = use_fp16

    def before_run(self, runner):
        runner.optimizer.zero_grad()

    def every_n_iters(self, runner, n):
        return (runner.iter + 1) % n == 0 if n > 0 else False

    def after_train_iter(self, runner):
        opt = runner.optimizer
        runner.outputs['loss'] /= self.update_interval
        boundary = self.every_n_iters(runner, self.update_interval)
        ctx = contextlib.nullcontext() if (boundary or not hasattr(opt, "no_sync")) else opt.no_sync()
        with ctx:
            runner.outputs['loss'].backward()
        if boundary:
            if self.grad_clip is not None:
                if hasattr(opt, "clip_grad_norm_"):
                    opt.clip_grad_norm_(**self.grad_clip)
                else:
                    torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], **self.grad_clip)
            opt.step()
            opt.zero_grad()


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
                if hasattr(m, "transformer"):
                    opt.attach_backbone(m)        # early buckets of the all-reduce start during backward
        return opt
    opt_cls = getattr(torch.optim, typ)
    return opt_cls(groups, **cfg)
