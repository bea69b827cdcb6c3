"""Two data-parallel ranks sharing the one MI355X of the test box (gloo carries the collective; under the
driver's multi-GPU launch the same code path runs on RCCL): the mean of the two ranks' flat adapter
gradients must equal the single-process gradient of the concatenated batch, parameters stay in lock-step
after a FlatAdamW step, and frozen tensors are never touched."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(T=2):
    import aim_amd
    from oracle import vit_clip_oracle as O
    cfg = dict(type='Recognizer3D',
               backbone=dict(type='ViT_CLIP', input_resolution=32, patch_size=16, num_frames=T, width=128, layers=2,
                             heads=2, drop_path_rate=0.0),
               cls_head=dict(type='I3DHead', in_channels=128, num_classes=5, dropout_ratio=0.0),
               test_cfg=dict(average_clips='prob'))
    torch.manual_seed(0)
    model = aim_amd.build_model(cfg)
    st = O.synth_state_dict(O.backbone_param_shapes(32, T, 16, 128, 2), seed=3)
    model.backbone.load_state_dict(st, strict=True)
    return model.cuda().train()


def _data(B=4, T=2):
    g = torch.Generator().manual_seed(77)
    return torch.randn((B, 1, 3, T, 32, 32), generator=g), torch.randint(0, 5, (B, 1), generator=g)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from aim_amd.dist import broadcast_module, build_optimizer, init_distributed
    init_distributed(backend="gloo")
    torch.cuda.set_device(0)
    model = _build()
    broadcast_module(model)
    opt = build_optimizer(model, dict(type='AdamW', lr=1e-2, weight_decay=0.05,
                                      paramwise_cfg=dict(custom_keys={'ln_post': dict(decay_mult=0.)})))
    imgs, label = _data()
    half = imgs.shape[0] // world
    sl = slice(rank * half, (rank + 1) * half)
    opt.zero_grad()
    losses = model(imgs[sl].cuda(), label[sl].cuda(), return_loss=True)
    losses["loss_cls"].backward()
    assert len(opt._buckets) == 2                    # layers 1 and 0 are early buckets of the flat buffer
    opt.all_reduce_grads()
    assert opt.early_launches == 2 and not opt._works      # both started from inside backward
    assert sorted(opt._reduced)[0][0] == 0 and opt._summed   # ... and the whole buffer has been through the collective
    # the collective is a SUM; the mean's 1 / world is folded into aim_adamw_flat(grad_scale=)
    grad = (opt.flat_g.detach() / world).cpu().clone()
    opt.step()
    after = opt.flat_p.detach().cpu().clone()
    # gradient accumulation: a micro-step under no_sync() starts no collective
    with opt.no_sync():
        model(imgs[sl].cuda(), label[sl].cuda(), return_loss=True)["loss_cls"].backward()
        opt.all_reduce_grads()
    assert opt.early_launches == 2
    # The reference's hook contract is `backward(); optimizer.step()` (mmaction/utils/optimizer.py:22-33): a step that is
    # NOT preceded by all_reduce_grads() must still finish the pending buckets and reduce the rest -- same parameters, bit
    # for bit, as the explicit order above.
    model2 = _build()
    broadcast_module(model2)
    opt2 = build_optimizer(model2, dict(type='AdamW', lr=1e-2, weight_decay=0.05,
                                        paramwise_cfg=dict(custom_keys={'ln_post': dict(decay_mult=0.)})))
    opt2.zero_grad()
    model2(imgs[sl].cuda(), label[sl].cuda(), return_loss=True)["loss_cls"].backward()
    assert opt2.early_launches == 2 and opt2._works and not opt2._summed      # two buckets in flight, nothing waited for
    assert opt2.grad_scale == 1.0
    opt2.step()
    assert opt2._summed and not opt2._works and opt2.grad_scale == 1.0 / world
    assert torch.equal(opt2.flat_p.detach().cpu(), after)
    # a second backward after the reduction has started would add local gradients to summed buckets: loud, never silent
    with pytest.raises(RuntimeError, match="no_sync"):
        model2(imgs[sl].cuda(), label[sl].cuda(), return_loss=True)["loss_cls"].backward()
    # DistOptimizerHook (update_interval = 2): the first micro-step communicates nothing, the boundary one reduces once;
    # clip_grad_norm_ sees the MEAN gradient
    from aim_amd.dist import DistOptimizerHook

    class _Runner:
        pass
    run = _Runner()
    run.optimizer, run.iter = opt2, 0
    hook = DistOptimizerHook(update_interval=2, grad_clip=dict(max_norm=1e9))
    hook.before_run(run)
    early0 = opt2.early_launches
    for it in range(2):
        run.iter = it
        run.outputs = dict(loss=model2(imgs[sl].cuda(), label[sl].cuda(), return_loss=True)["loss_cls"])
        hook.after_train_iter(run)
        assert opt2.early_launches == early0 + (2 if it == 1 else 0)
    assert opt2.step_count == 2 and float(opt2.flat_g.abs().max()) == 0.0       # stepped once more, then zero_grad
    q.put((rank, grad.numpy(), after.numpy()))      # by value: the worker exits before the parent reads
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    (_, g0, a0), (_, g1, a1) = [(r, torch.from_numpy(g), torch.from_numpy(a)) for r, g, a in res]
    assert torch.equal(g0, g1) and torch.equal(a0, a1)              # ranks hold identical reduced grads / params
    # single-process reference on the full batch
    from aim_amd.dist import build_optimizer
    model = _build()
    frozen_before = {n: p.detach().clone() for n, p in model.named_parameters() if not p.requires_grad}
    opt = build_optimizer(model, dict(type='AdamW', lr=1e-2, weight_decay=0.05,
                                      paramwise_cfg=dict(custom_keys={'ln_post': dict(decay_mult=0.)})))
    imgs, label = _data()
    opt.zero_grad()
    model(imgs.cuda(), label.cuda(), return_loss=True)["loss_cls"].backward()
    ref = opt.flat_g.detach().cpu()
    rel = ((g0 - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel          # bf16 kernels, different batch tiling: not bitwise
    opt.step()
    assert all(torch.equal(p.detach(), frozen_before[n]) for n, p in model.named_parameters() if n in frozen_before)
    # the two-rank parameters after one step == the single-process ones (same mean gradient, same AdamW)
    relp = ((a0 - opt.flat_p.detach().cpu()).norm() / a0.norm()).item()
    assert relp < 1e-3, relp


def test_flat_adamw_state_dict_is_torch_compatible():
    """FlatAdamW.state_dict() loads into torch.optim.AdamW over the same parameter order (and back): checkpoints
    written by mmcv's hook (mmcv_custom/runner/checkpoint.py:39-80 stores optimizer.state_dict() as is) interchange."""
    from aim_amd.dist import build_optimizer
    cfg = dict(type='AdamW', lr=1e-2, weight_decay=0.05, paramwise_cfg=dict(custom_keys={'ln_post': dict(decay_mult=0.)}))
    imgs, label = _data()
    model = _build()
    opt = build_optimizer(model, cfg)
    for _ in range(2):
        opt.zero_grad()
        model(imgs.cuda(), label.cuda(), return_loss=True)["loss_cls"].backward()
        opt.step()
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and all({"step", "exp_avg", "exp_avg_sq"} == set(v) for v in sd["state"].values())
    # torch's AdamW over the same parameters in the same group order accepts it and continues identically
    groups = [dict(params=list(g["params"]), lr=g["lr"], weight_decay=g["weight_decay"]) for g in opt.param_groups]
    topt = torch.optim.AdamW(groups, lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    topt.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    fake = torch.randn(opt.flat_g.shape, generator=g).cuda() * 1e-3
    before = opt.flat_p.detach().clone()
    opt.flat_g.copy_(fake)
    opt.step()
    after_flat = opt.flat_p.detach().clone()
    opt.flat_p.copy_(before)                # params are views of flat_p: rewind, then let torch step from the same state
    opt.flat_g.copy_(fake)
    topt.step()
    for p, off, n in opt._views:          # (the padding slots between tensors belong to nobody)
        d = (opt.flat_p[off:off + n] - after_flat[off:off + n]).abs().max().item()
        assert d < 1e-6, (off, n, d)
    # and back
    model2 = _build()
    opt2 = build_optimizer(model2, cfg)
    opt2.load_state_dict(topt.state_dict())
    assert opt2.step_count == 3
    a, b = opt2.state_dict()["state"], topt.state_dict()["state"]
    assert all(torch.equal(a[k]["exp_avg"], b[k]["exp_avg"]) and torch.equal(a[k]["exp_avg_sq"], b[k]["exp_avg_sq"]) for k in b)


def test_gradient_accumulation_matches_full_batch():
    """DistOptimizerHook micro-batching (mmaction/utils/optimizer.py:22-33): loss /= update_interval per
    micro-step, one optimizer step (and one all-reduce) per interval."""
    from aim_amd.dist import build_optimizer
    imgs, label = _data()
    grads = []
    for k in (1, 2):
        model = _build()
        opt = build_optimizer(model, dict(type='AdamW', lr=1e-2, weight_decay=0.0))
        opt.zero_grad()
        mb = imgs.shape[0] // k
        for i in range(k):
            sl = slice(i * mb, (i + 1) * mb)
            loss = model(imgs[sl].cuda(), label[sl].cuda(), return_loss=True)["loss_cls"] / k
            loss.backward()
        grads.append(opt.flat_g.detach().clone())
    rel = ((grads[0] - grads[1]).norm() / grads[0].norm()).item()
    assert rel < 2e-2, rel


def _rccl_worker(port, q):
    """Single-rank RCCL group on the one GPU of the test box: the REAL `nccl` backend (RCCL) runs the optimizer's call
    pattern -- async bucket all-reduces issued from inside backward on a communication stream, the tail reduce, the waits,
    the packed log all-reduce -- and must leave the gradients of a world-size-1 mean untouched."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    from aim_amd.dist import broadcast_module, build_optimizer
    from aim_amd.recognizer import Recognizer3D
    model = _build()
    broadcast_module(model)
    opt = build_optimizer(model, dict(type='AdamW', lr=1e-2, weight_decay=0.05))
    imgs, label = _data()
    # reference gradient without collectives
    opt.zero_grad()
    model(imgs.cuda(), label.cuda(), return_loss=True)["loss_cls"].backward()
    opt.all_reduce_grads()
    ref = opt.flat_g.detach().clone()
    assert opt.early_launches == 0
    opt.force_collectives = True
    diff = None
    for it in range(3):
        opt.zero_grad()
        losses = model(imgs.cuda(), label.cuda(), return_loss=True)
        losses["loss_cls"].backward()
        opt.all_reduce_grads()
        loss, log_vars = Recognizer3D._parse_losses(losses)
        if it == 0:          # same parameters, same data: a 1-rank SUM must reproduce the un-reduced gradient bit for bit
            diff = float((opt.flat_g.detach() - ref).abs().max())
        opt.step()
    torch.cuda.synchronize()
    q.put((opt.early_launches, diff, float(log_vars["loss_cls"])))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_call_pattern_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    early, diff, loss = q.get(timeout=240)
    p.join(60)
    assert p.exitcode == 0
    assert early == 6 and loss == loss        # two early buckets per step x 3 steps, through RCCL
    assert diff == 0.0
