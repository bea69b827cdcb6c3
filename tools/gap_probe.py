#!/usr/bin/env python3
"""Launch gaps between consecutive kernels of the critical (main-stream) chain in a rocprofv3 kernel trace.

usage: gap_probe.py <run_results.db>
The trace's stream ids do not separate the HIP streams reliably, so the chain is picked by kernel name: the large GEMMs,
LayerNorm and attention kernels run on the main stream in dependency order.  Caveat: the forward's class-token chain launches
the SAME ln_fwd kernel on its 512 class rows on the side stream, so "ln_fwd -> X" pairs mix in side-stream launches (the
negative ln_fwd -> ln_fwd entry is that overlap); tools/chain_probe.py shows the true per-block timeline."""
import collections
import re
import sqlite3
import sys

MAIN = re.compile(r"gemm256_kernel|ln_fwd_kernel|ln_bwd_kernel|ln_bwd_fsum|[^s]_attn_fwd_kernel|^void \(anonymous namespace\)::attn_fwd|::attn_bwd_|embed_ln|embed_bwd_kernel|patchify")


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = [(a, b, n) for a, b, n in cur.execute(f"select d.start, d.end, s.display_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start")
            if MAIN.search(n)]
    gaps = collections.defaultdict(lambda: [0, 0.0])
    tot = 0.0
    n = 0
    for (a0, b0, n0), (a1, b1, n1) in zip(rows, rows[1:]):
        g = (a1 - b0) / 1e3
        if g > 2000:          # step boundary
            continue
        short = lambda x: re.sub(r"\(.*", "", x.replace("void ", "").replace("(anonymous namespace)::", ""))[:28]
        key = short(n0) + " -> " + short(n1)
        gaps[key][0] += 1
        gaps[key][1] += g
        tot += g
        n += 1
    print(f"{n} hand-overs, {tot / 1e3:.2f} ms total gap (all steps in the trace), mean {tot / max(n, 1):.1f} us")
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"  {k:60s} n={v[0]:4d} mean {v[1] / v[0]:7.1f} us  total {v[1] / 1e3:7.2f} ms")


if __name__ == "__main__":
    main()
