"""world_size-2 data-parallel plumbing on CPU (gloo): the flat adapter-gradient all-reduce, the initial
broadcast, the sampler partition and the coalesced log all-reduce.  The GPU step itself is covered by
tests/test_backbone_gpu.py; the collective call pattern is identical under RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from aim_amd.dist import FlatGradReducer, broadcast_module, init_distributed, shard_indices
    from aim_amd.recognizer import Recognizer3D
    r, _, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # ranks start with DIFFERENT weights
    lin = torch.nn.Sequential(torch.nn.Linear(8, 4), torch.nn.Linear(4, 3))
    lin[0].weight.requires_grad_(False)                # frozen tensors are never communicated
    broadcast_module(lin)
    red = FlatGradReducer(lin.parameters())
    assert red.numel == 4 + 4 * 3 + 3
    x = torch.full((2, 8), float(rank + 1))
    red.zero_grad()
    lin(x).sum().backward()
    local = red.flat.clone()
    red.all_reduce()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(red.flat, sum(gathered) / world, atol=1e-6)
    assert lin[0].weight.grad is None
    assert lin[1].weight.grad.data_ptr() >= red.flat.data_ptr()       # grads are views of the flat buffer
    # parameters identical after broadcast
    w0 = [torch.zeros_like(lin[1].weight) for _ in range(world)]
    dist.all_gather(w0, lin[1].weight.data)
    assert torch.equal(w0[0], w0[1])
    # sampler partition: disjoint, padded to a multiple of world
    idx = shard_indices(11, rank, world, seed=3, epoch=2)
    allidx = [None] * world
    dist.all_gather_object(allidx, idx)
    assert len(idx) == 6 and sorted(set(sum(allidx, []))) == list(range(11))
    # coalesced log all-reduce (one call instead of four, recognizers/base.py:237-242)
    losses = dict(loss_cls=torch.tensor(float(rank + 1)), top1_acc=torch.tensor(0.5 * rank), top5_acc=torch.tensor(1.0))
    loss, log_vars = Recognizer3D._parse_losses(losses)
    assert abs(log_vars["loss_cls"] - 1.5) < 1e-6 and abs(log_vars["top1_acc"] - 0.25) < 1e-6
    assert abs(log_vars["loss"] - 1.5) < 1e-6 and float(loss) == float(rank + 1)
    q.put(rank)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_dp():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def test_shard_indices_single_rank():
    from aim_amd.dist import shard_indices
    assert shard_indices(5, 0, 1, shuffle=False) == [0, 1, 2, 3, 4]
    a, b = shard_indices(5, 0, 2, shuffle=False), shard_indices(5, 1, 2, shuffle=False)
    assert a == [0, 2, 4] and b == [1, 3, 0]            # padded by wrapping, like DistributedSampler


def test_config_and_registry_surface():
    """Config inheritance / dotted overrides / registry build errors (host logic, no GPU)."""
    import aim_amd
    cfg = aim_amd.Config.fromfile(os.path.join(os.path.dirname(__file__), "data", "vitclip_tiny_cfg.py"))
    assert cfg.model.type == "Recognizer3D" and cfg.model.backbone.num_frames == 4
    assert cfg.model.backbone.input_resolution == 32                      # inherited from _base_
    assert cfg.model.cls_head.num_classes == 7 and cfg.model.test_cfg == dict(average_clips="prob", max_testing_views=4)
    cfg.merge_from_dict({"model.backbone.drop_path_rate": 0.3, "optimizer.lr": 1e-3})
    assert cfg.model.backbone.drop_path_rate == 0.3 and cfg.optimizer.lr == 1e-3
    model = aim_amd.build_model(cfg.model)
    assert isinstance(model.backbone, aim_amd.ViT_CLIP) and isinstance(model.cls_head, aim_amd.I3DHead)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert len(names) == 2 * 12 + 3 + 2 and all(("Adapter" in n or "ln_post" in n or "temporal" in n or "cls_head" in n)
                                                 for n in names)
    # D_fc2 zero-init: the model starts equal to the frozen CLIP (vit_clip.py:386-411)
    assert all(float(p.abs().max()) == 0 for n, p in model.named_parameters() if "D_fc2" in n)
    with pytest.raises(KeyError, match="not in the models registry"):
        aim_amd.build_backbone(dict(type="NoSuchBackbone"))
    with pytest.raises(NotImplementedError):
        aim_amd.ViT_CLIP(32, 2, 16, 128, 1, 2, 0.0, shift=True)
    with pytest.raises(TypeError, match="pretrained must be a str or None"):
        aim_amd.ViT_CLIP(32, 2, 16, 128, 1, 2, 0.0, pretrained=3).init_weights()
    from aim_amd.dist import build_optimizer
    opt = build_optimizer(model, cfg.optimizer.to_dict() if hasattr(cfg.optimizer, "to_dict") else dict(cfg.optimizer))
    wd = {id(p): g["weight_decay"] for g in opt.param_groups for p in g["params"]}
    named = dict(model.named_parameters())
    assert wd[id(named["backbone.ln_post.weight"])] == 0.0
    assert wd[id(named["backbone.transformer.resblocks.0.S_Adapter.D_fc1.weight"])] == 0.05


def test_top_k_accuracy_device_matches_numpy():
    import numpy as np
    from aim_amd.recognizer import top_k_accuracy, top_k_accuracy_device
    g = torch.Generator().manual_seed(0)
    s = torch.randn(64, 20, generator=g)
    lab = torch.randint(0, 20, (64,), generator=g)
    ref = top_k_accuracy(s.numpy(), lab.numpy(), (1, 5))
    got = [float(v) for v in top_k_accuracy_device(s, lab, (1, 5))]
    assert np.allclose(ref, got)


def test_lazy_log_vars_and_aim_registry_surface():
    """Host logic added in round 2 (no GPU): `_parse_losses` hands back lazily materialised floats; the stock-AIM class is
    reachable through the registry with the reference's constructor keywords (vitclip_aim.py:356-358) and refuses the
    branches that are not built."""
    import aim_amd
    from aim_amd.recognizer import LazyLogVars, Recognizer3D
    losses = dict(loss_cls=torch.tensor(2.5), top1_acc=torch.tensor(0.25), top5_acc=torch.tensor(0.75))
    loss, lv = Recognizer3D._parse_losses(losses)
    assert isinstance(lv, LazyLogVars) and list(lv) == ["loss_cls", "top1_acc", "top5_acc", "loss"]
    assert float(loss) == 2.5 and lv["loss"] == 2.5 and lv.get("top1_acc") == 0.25
    assert dict(lv.items()) == dict(loss_cls=2.5, top1_acc=0.25, top5_acc=0.75, loss=2.5)
    m = aim_amd.build_backbone(dict(type='AIM', input_resolution=32, num_frames=2, patch_size=16, width=128, layers=2, heads=2,
                                    drop_path_rate=0.1, num_tadapter=1, adapter_scale=0.5, pretrained=None, prompt=True,
                                    wind_attn=False, window_size=(32, 2, 2), not_shift=True))
    assert isinstance(m, aim_amd.AIM) and isinstance(m, aim_amd.ViT_CLIP) and m.variant == 'aim'
    m.init_weights()
    names = sorted(n for n, p in m.named_parameters() if p.requires_grad)
    ref = aim_amd.ViT_CLIP(32, 2, 16, 128, 2, 2, 0.1)
    ref.init_weights()
    assert names == sorted(n for n, p in ref.named_parameters() if p.requires_grad)        # same state_dict keys / freeze policy
    with pytest.raises(NotImplementedError, match="wind_attn"):
        aim_amd.AIM(32, 2, 16, 128, 2, 2, 0.1, wind_attn=True)
    with pytest.raises(NotImplementedError, match="num_tadapter"):
        aim_amd.AIM(32, 2, 16, 128, 2, 2, 0.1, num_tadapter=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 2, 32, 32))


def test_dist_optimizer_hook_micro_stepping():
    """DistOptimizerHook (mmaction/utils/optimizer.py:9-33) on a recording optimizer: loss /= update_interval, backward on
    every iteration -- inside no_sync() on the non-boundary ones --, clip + step + zero_grad on every update_interval-th."""
    import contextlib

    from aim_amd.dist import DistOptimizerHook

    class Opt:
        def __init__(self):
            self.log, self.in_no_sync = [], False

        @contextlib.contextmanager
        def no_sync(self):
            self.in_no_sync = True
            try:
                yield
            finally:
                self.in_no_sync = False

        def zero_grad(self):
            self.log.append("zero")

        def step(self):
            self.log.append("step")

        def clip_grad_norm_(self, max_norm, norm_type=2.0):
            self.log.append(("clip", max_norm))

    class Loss:
        def __init__(self, opt, v):
            self.opt, self.v = opt, v

        def __itruediv__(self, k):
            self.v /= k
            return self

        def backward(self):
            self.opt.log.append(("backward", self.v, self.opt.in_no_sync))

    class Runner:
        pass

    opt = Opt()
    run = Runner()
    run.optimizer = opt
    hook = DistOptimizerHook(update_interval=3, grad_clip=dict(max_norm=40), coalesce=True, bucket_size_mb=-1, use_fp16=True)
    hook.before_run(run)
    for it in range(6):
        run.iter = it
        run.outputs = dict(loss=Loss(opt, 3.0))
        hook.after_train_iter(run)
    micro = [("backward", 1.0, True), ("backward", 1.0, True), ("backward", 1.0, False), ("clip", 40), "step", "zero"]
    assert opt.log == ["zero"] + micro + micro
