#!/usr/bin/env python3
"""Idle gap between dependent kernels of one stream: plain, with an event record between them, with a wait on an
event another stream recorded long ago, after a kernel that leaves many dirty lines in the L2s."""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import ops  # noqa: E402,F401

dev = "cuda"
small = torch.zeros(4096, device=dev)
big_src = torch.randn(64 * 1024 * 1024, device=dev)          # 256 MB
big_dst = torch.empty_like(big_src)
side = torch.cuda.Stream()
N = 400


def timed(fn, label, per=1):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:58s} {e0.elapsed_time(e1) * 1e3 / (N * per):7.2f} us per kernel   (host {1e6 * (time.perf_counter() - t0) / (N * per):6.2f} us)", flush=True)


def plain():
    for _ in range(N):
        small.add_(1.0)


def with_record():
    for _ in range(N):
        small.add_(1.0)
        ev = torch.cuda.Event()
        ev.record()


old = torch.cuda.Event()
with torch.cuda.stream(side):
    small.clone()
    old.record()
torch.cuda.synchronize()


def with_wait_old():
    for _ in range(N):
        small.add_(1.0)
        torch.cuda.current_stream().wait_event(old)


def fork_join():
    main = torch.cuda.current_stream()
    for _ in range(N):
        small.add_(1.0)
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ev2 = torch.cuda.Event()
            ev2.record()
        main.wait_event(ev2)


def big_then_small():
    for _ in range(N // 8):
        big_dst.copy_(big_src)          # 256 MB written: dirty lines in every L2 at kernel end
        small.add_(1.0)


def big_only():
    for _ in range(N // 8):
        big_dst.copy_(big_src)


timed(plain, "tiny dependent kernels back to back")
timed(with_record, "... with an event record after each")
timed(with_wait_old, "... with a wait on an old event of another stream")
timed(fork_join, "... with record -> side-stream wait+record -> wait (empty fork/join)")
N8 = N
timed(big_only, "256 MB copy kernels back to back", per=1 / 8)
timed(big_then_small, "256 MB copy + tiny kernel (pair)", per=1 / 8)
