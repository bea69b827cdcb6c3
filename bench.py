#!/usr/bin/env python3
"""bench.py -- clips/s of the AIM ViT-CLIP+Adapter training step (fwd + bwd + AdamW) on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched by
``python -m torch.distributed.run --nproc-per-node N ...`` (one rank per GPU, RCCL).  Rank 0 prints ONE
JSON line.  Workload = BASELINE.json configs[1]: ViT-B/16 + AIM, 8 frames 224^2, bf16 operands / fp32
accumulate, 64 clips per GPU, synthetic N(0,1) clips and U{0..399} labels, random-init weights
(pretrained=None, D_fc2 ~ N(0, .02) so the adapters are live), drop_path 0.2 and head dropout 0.5 ON.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.6        # 256 CU x 2.4 GHz x 4096 flop/clk/CU, dense (MI355X_MICROARCH.md: ~2.5 PF)
GF_PER_CLIP = {"fwd": 293.1, "bwd": 314.0}   # SURVEY section 8(d), ViT-B/16 T=8, algorithmic
GF_PER_CLIP_L14_T16 = 5608.5                 # SURVEY section 8(d), ViT-L/14 T=16 fwd+bwd, algorithmic
GF_PER_VIEW_L14_T32 = 5405.1                 # SURVEY section 8(d), ViT-L/14 T=32 forward (inference), algorithmic
PEAK_FP8_TFLOPS = 2 * PEAK_BF16_TFLOPS       # dense fp8 MFMA (block-scaled K=128 form): 2x bf16 (MI355X_MICROARCH.md: ~5 PF)
ARCH = {"B16": dict(patch_size=16, width=768, layers=12, heads=12),
        "L14": dict(patch_size=14, width=1024, layers=24, heads=16)}     # configs/recognition/vit/vitclip_{base,large}_k400.py


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the ViT-L/14 16-frame block (BASELINE configs[3] shape)")
    ap.add_argument("--no-inference", action="store_true", help="skip the fp8 multi-view inference block (BASELINE configs[4] shape)")
    ap.add_argument("--inference-only", action="store_true", help="profiling aid: run only the inference block and print it")
    ap.add_argument("--secondary-only", action="store_true", help="profiling aid: run only the ViT-L/14 training block and print it")
    return ap.parse_args()


def build_model(frames, dev, arch="B16"):
    import aim_amd
    a = ARCH[arch]
    cfg = dict(
        type='Recognizer3D',
        backbone=dict(type='ViT_CLIP', input_resolution=224, num_frames=frames, drop_path_rate=0.2, adapter_scale=0.5,
                      pretrained=None, **a),
        cls_head=dict(type='I3DHead', in_channels=a["width"], num_classes=400, spatial_type='avg', dropout_ratio=0.5),
        test_cfg=dict(average_clips='prob'))
    torch.manual_seed(0)
    model = aim_amd.build_model(cfg)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "D_fc2" in n:
                p.normal_(0, 0.02)
    return model.to(dev).train()


def cpu_baseline(frames):
    """The CPU oracle (literal restatement of the reference forward, pinned by tests/golden) timed on the
    host cores: 1 clip, fp32, fwd + bwd of backbone + head + CE.  Bounded sample (~10-30 s)."""
    from oracle import vit_clip_oracle as O
    # every core this process may run on (affinity mask, capped by a cgroup CPU quota when one is set)
    logical = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = logical
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    threads = max(1, min(allowed, quota) if quota else allowed)
    torch.set_num_threads(threads)
    st = O.synth_state_dict(O.backbone_param_shapes(224, frames, 16, 768, 12), seed=0)
    names = O.trainable_names(st)
    for n in names:
        st[n].requires_grad_(True)
    fc_w = (torch.randn(400, 768) * 0.01).requires_grad_(True)
    fc_b = torch.zeros(400, requires_grad=True)
    imgs = torch.randn(1, 3, frames, 224, 224)
    label = torch.tensor([7])

    def step():
        y = O.ref_backbone(imgs, st, 12, frames)
        loss = O.ref_cross_entropy(O.ref_i3d_head(y, fc_w, fc_b), label)
        torch.autograd.grad(loss, [st[n] for n in names] + [fc_w, fc_b])

    step()
    times = []
    t_end = time.time() + 20.0
    while len(times) < 5 or (time.time() < t_end and len(times) < 8):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    med = sorted(times)[len(times) // 2]
    # BASELINE configs[0]: ViT-B/16 + AIM, 2 frames, batch 1, fp32 forward (the reference's own CPU-runnable case)
    st1 = O.synth_state_dict(O.backbone_param_shapes(224, 2, 16, 768, 12), seed=0)
    imgs1 = torch.randn(1, 3, 2, 224, 224)
    with torch.no_grad():
        O.ref_backbone(imgs1, st1, 12, 2)
        t1 = []
        for _ in range(5):
            t0 = time.time()
            O.ref_backbone(imgs1, st1, 12, 2)
            t1.append(time.time() - t0)
    cfg1 = sorted(t1)[2]
    model_name = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=round(1.0 / med, 4), unit="clips/s", cores=threads, kind="port",
                sample=f"1 clip x {len(times)} steps (median), ViT-B/16 T={frames} fp32 fwd+bwd+head on {model_name} "
                       f"({logical} logical CPUs, {allowed} in the affinity mask, cgroup quota {quota}; {threads} threads used)",
                cfg1_forward_s=round(cfg1, 4),
                cfg1_sample="BASELINE configs[0]: ViT-B/16 + AIM, 2 frames 224^2, batch 1, fp32 forward, median of 5")


def secondary_l14(dev, rank, world, steps=3, warmup=1, clips=32, frames=16):
    """BASELINE configs[3] per-GPU shape: ViT-L/14 + AIM, 16 frames, 32 clips per GPU, the same train step
    (fwd + bwd + adapter-grad all-reduce + AdamW), timed like the primary (barrier + synchronize, max over ranks)."""
    from aim_amd.dist import broadcast_module, build_optimizer
    model = build_model(frames, dev, "L14")
    broadcast_module(model)
    opt = build_optimizer(model, dict(type='AdamW', lr=3e-4, betas=(0.9, 0.999), weight_decay=0.05))
    g = torch.Generator(device="cpu").manual_seed(4321 + rank)
    imgs = torch.randn((clips, 1, 3, frames, 224, 224), generator=g).to(dev)
    label = torch.randint(0, 400, (clips, 1), generator=g).to(dev)

    def step():
        opt.zero_grad()
        loss = model(imgs, label, return_loss=True)["loss_cls"]
        loss.backward()
        opt.step()                  # (finishes the gradient reduction the backward started: dist.FlatAdamW.step)
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    value = clips * world * steps / dt
    return {"workload": "BASELINE configs[3] per-GPU shape: ViT-L/14 + AIM adapters, 16 frames 224^2, 32 clips/GPU, "
                        "fwd+bwd+AdamW, drop_path 0.2, head dropout 0.5, synthetic",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 2), "global_batch": clips * world,
            "mfma_frac_whole_step": round(value / world * GF_PER_CLIP_L14_T16 / 1e3 / PEAK_BF16_TFLOPS, 4),
            "loss": round(float(loss.detach()), 4),
            "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)}


def inference_l14(dev, rank, world, samples=4, views=3, frames=32, steps=4, warmup=2):
    """BASELINE configs[4] per-GPU shape: ViT-L/14 + AIM, 32 frames, 3 views per sample (3-crop), multi-view test path
    ``Recognizer3D.forward_test`` -> ``average_clip('prob')``, fp8 e4m3 GEMM operands; the bf16 path is timed beside it
    in the same process.  Replicas only: inference shards samples over ranks with no collective."""
    import aim_amd
    a = ARCH["L14"]
    cfg = dict(type='Recognizer3D',
               backbone=dict(type='ViT_CLIP', input_resolution=224, num_frames=frames, drop_path_rate=0.2, adapter_scale=0.5,
                             pretrained=None, **a),
               cls_head=dict(type='I3DHead', in_channels=a["width"], num_classes=400, spatial_type='avg', dropout_ratio=0.5),
               test_cfg=dict(average_clips='prob'))
    torch.manual_seed(0)
    model = aim_amd.build_model(cfg)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "D_fc2" in n:
                p.normal_(0, 0.02)
    model = model.to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(999 + rank)
    imgs = torch.randn((samples, views, 3, frames, 224, 224), generator=g).to(dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out = {}
    for prec in ("bf16", "fp8"):
        model.backbone.set_inference_precision(prec)
        with torch.no_grad():
            for _ in range(warmup):
                model._do_test(imgs)
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                probs = model._do_test(imgs)
            fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        out[prec] = dict(views_per_s=round(samples * views * world * steps / dt, 2), ms_per_step=round(dt / steps * 1e3, 2),
                         pred=probs.argmax(1).tolist())
    v8 = out["fp8"]["views_per_s"]
    return {"workload": f"BASELINE configs[4] per-GPU shape: ViT-L/14 + AIM, {frames} frames 224^2, {views} views x {samples} samples "
                        "per step, multi-view inference (forward_test, average_clips='prob'), fp8 e4m3 GEMM operands / fp32 accumulate",
            "value": v8, "unit": "views/s", "n_gpus": world, "steps": steps, "warmup": warmup, "dtype": "fp8",
            "ms_per_step": out["fp8"]["ms_per_step"],
            "fp8_mfma_frac": round(v8 / world * GF_PER_VIEW_L14_T32 / 1e3 / PEAK_FP8_TFLOPS, 4),
            "bf16_same_process": {"views_per_s": out["bf16"]["views_per_s"], "ms_per_step": out["bf16"]["ms_per_step"],
                                  "bf16_mfma_frac": round(out["bf16"]["views_per_s"] / world * GF_PER_VIEW_L14_T32 / 1e3 / PEAK_BF16_TFLOPS, 4)},
            "top1_agreement_fp8_vs_bf16": sum(int(a_ == b_) for a_, b_ in zip(out["fp8"]["pred"], out["bf16"]["pred"])) / samples,
            "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)}


def main():
    args = parse()
    from aim_amd.dist import broadcast_module, build_optimizer, init_distributed
    from aim_amd import ops
    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.inference_only:
        print(json.dumps(inference_l14(dev, rank, world)), flush=True)
        if os.environ.get("AIM_JOIN_STATS") and rank == 0:
            from aim_amd import backbone as _bb
            print("join stalls, ms per 12 steps (6 bf16 + 6 fp8):", {k: round(v, 3) for k, v in _bb.join_stats().items()}, file=sys.stderr)
        return
    if args.secondary_only:
        print(json.dumps(secondary_l14(dev, rank, world)), flush=True)
        if os.environ.get("AIM_JOIN_STATS") and rank == 0:
            from aim_amd import backbone as _bb
            print("join stalls, ms per step:", {k: round(v / 4, 3) for k, v in _bb.join_stats().items()}, file=sys.stderr)
        return
    model = build_model(args.frames, dev)
    broadcast_module(model)
    opt = build_optimizer(model, dict(
        type='AdamW', lr=3e-4, betas=(0.9, 0.999), weight_decay=0.05,
        paramwise_cfg=dict(custom_keys={k: dict(decay_mult=0.) for k in
                                        ('class_embedding', 'positional_embedding', 'ln_1', 'ln_2', 'ln_pre', 'ln_post')})))

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    B = args.batch
    imgs = torch.randn((B, 1, 3, args.frames, 224, 224), generator=g).to(dev)      # resident in HBM
    label = torch.randint(0, 400, (B, 1), generator=g).to(dev)

    def step():
        opt.zero_grad()
        losses = model(imgs, label, return_loss=True)
        loss = losses["loss_cls"]
        loss.backward()
        opt.step()                  # the reference's hook contract (mmaction/utils/optimizer.py:22-33): step() finishes the
        return losses               # adapter-gradient all-reduce whose first two buckets started inside backward

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    timing = (not args.no_kernel_timing) and rank == 0
    if timing:
        ops.GEMM_TIMER.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step()
    fence()
    dt = time.perf_counter() - t0
    ops.GEMM_TIMER.stop()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_val = float(losses["loss_cls"].detach())
    if os.environ.get("AIM_JOIN_STATS") and rank == 0:
        from aim_amd import backbone as _bb
        print("join stalls, ms per step:", {k: round(v / (args.steps + args.warmup), 3) for k, v in _bb.join_stats().items()}, file=sys.stderr)

    if rank == 0:
        clips = B * world * args.steps
        value = clips / dt
        gf = GF_PER_CLIP["fwd"] + GF_PER_CLIP["bwd"]
        out = {
            "metric": "clips/sec (8-frame 224^2 ViT-B/16 AIM fwd+bwd)", "value": round(value, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ViT-B/16 + AIM adapters, 8 frames 224^2, 64 clips/GPU, "
                                   "K400-shaped synthetic clips (400 random labels), fwd+bwd+AdamW",
                       "global_batch": B * world, "frames": args.frames, "parallelism": f"dp{world}",
                       "drop_path_rate": 0.2, "head_dropout": 0.5},
            "mfma_frac_whole_step": round(value / world * gf / 1e3 / PEAK_BF16_TFLOPS, 4) if args.frames == 8 else None,
            "loss": round(loss_val, 4),
        }
        roof = None
        if timing:
            # launches of >= 10 GFLOP: all on the persistent 256x256 kernel (gemm256.hip).  They are timed where they run: in
            # the backward they share the chip with the detached weight-gradient stream (Streams)
            names = {0: "gemm256_kernel<EPI_BF16>", 1: "gemm256_kernel<EPI_ACT>", 2: "gemm256_kernel<EPI_DACT>",
                     3: "gemm256_kernel<EPI_F32>", 4: "gemm256_kernel<EPI_EXPSUM>"}
            summ = ops.GEMM_TIMER.summary()
            if summ:
                tot_ms = sum(d["ms"] for d in summ.values())
                tot_fl = sum(d["flops"] for d in summ.values())
                dom = max(summ, key=lambda k: summ[k]["ms"])
                d = summ[dom]
                ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
                # HBM-side bytes per launch of that kernel: NOT measured in this run -- read from the committed PMC
                # passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command, profiles/)
                traffic, traffic_src = None, None
                mfma_busy = None
                for name in ("r03_gemm_pmc.json", "r02_gemm_pmc.json", "r01_gemm_traffic.json"):
                    try:
                        tj = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]["gemm256_kernel<%d>" % dom]
                        traffic = round(tj["hbm_bytes_per_launch"])
                        mfma_busy = tj.get("mfma_busy_frac")
                        traffic_src = "profiles/" + name + " (committed rocprofv3 PMC passes of this command, not measured in this run)"
                        break
                    except Exception:
                        pass
                roof = {"bound": "mfma", "kernel": names[dom], "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                        "traffic_source": traffic_src,
                        "mfma_busy_frac_pmc": None if mfma_busy is None else round(mfma_busy, 4),
                        "launches_per_step": d["launches"] / args.steps,
                        "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                        "flops_per_launch": d["flops"] / d["launches"],
                        "all_large_gemms": {"TFLOP/s": round(tot_fl / (tot_ms * 1e-3) / 1e12, 1),
                                            "ms_per_step": round(tot_ms / args.steps, 3),
                                            "share_of_step": round(tot_ms / (dt * 1e3), 3)},
                        "per_variant": {names[k]: {"TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                                                   "avg_ms": round(v["ms"] / v["launches"], 4),
                                                   "launches_per_step": v["launches"] / args.steps}
                                        for k, v in sorted(summ.items())}}
        out["roofline"] = roof
    # second shape (after the primary timed region; every rank takes part)
    sec = None
    del model, opt, imgs, label, losses
    torch.cuda.empty_cache()
    if not args.no_secondary and args.frames == 8:
        torch.cuda.reset_peak_memory_stats(dev)
        try:
            sec = secondary_l14(dev, rank, world)
        except Exception as e:          # the primary line must still be printed ...
            if world > 1:               # ... but never by leaving the other ranks blocked inside a collective
                raise
            sec = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
    inf = None
    if not args.no_inference and args.frames == 8:
        torch.cuda.reset_peak_memory_stats(dev)
        try:
            inf = inference_l14(dev, rank, world)
        except Exception as e:
            if world > 1:
                raise
            inf = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
    if rank == 0:
        out["secondary"] = sec
        out["inference"] = inf
        out["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(args.frames)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
