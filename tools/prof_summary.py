#!/usr/bin/env python3
"""rocprofv3 results database -> markdown summary (per-kernel table, per-stream busy time).

usage: prof_summary.py <run_results.db> <steps-in-trace> [title]
The database comes from  rocprofv3 --kernel-trace --stats -d DIR -o run -- python3 bench.py ...  (kernel dispatch table)."""
import collections
import re
import sqlite3
import sys


def main():
    db, steps = sys.argv[1], int(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --stats"
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(cur.execute(f"select d.stream_id, d.start, d.end, s.display_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))

    def short(n):
        n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
        n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n)
        return n[:72]

    busy = collections.defaultdict(float)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for st, a, b, n in rows:
        busy[st] += (b - a) / 1e6
        agg[short(n)][0] += 1
        agg[short(n)][1] += (b - a) / 1e6
    tot = sum(v[1] for v in agg.values())
    print(f"# {title}\n")
    print(f"{len(rows)} kernel dispatches over {steps} steps; kernel-time sum {tot / steps:.2f} ms/step "
          f"(streams overlap, so the sum exceeds wall time).\n")
    print("Busy time per HIP stream (ms/step): " + ", ".join(f"stream {k}: {v / steps:.2f}" for k, v in sorted(busy.items())) + "\n")
    print("| kernel | calls/step | ms/step | avg us | % of kernel time |\n|---|---|---|---|---|")
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:48]:
        print(f"| `{n}` | {v[0] / steps:.1f} | {v[1] / steps:.3f} | {v[1] / v[0] * 1e3:.1f} | {100 * v[1] / tot:.2f} |")
    if len(sys.argv) > 4 and sys.argv[4] == "main":
        main_stream(rows, steps, short)


def main_stream(rows, steps, short):
    """Main stream only (the stream with the most busy time): kernel time by name, and the idle gaps between its kernels."""
    import collections
    busy = collections.defaultdict(float)
    for st, a, b, n in rows:
        busy[st] += b - a
    ms = max(busy, key=busy.get)
    r = [(a, b, short(n)) for st, a, b, n in rows if st == ms]
    agg = collections.defaultdict(lambda: [0, 0.0])
    gaps = collections.defaultdict(lambda: [0, 0.0])
    for i, (a, b, n) in enumerate(r):
        agg[n][0] += 1
        agg[n][1] += (b - a) / 1e6
        if i:
            g = (a - r[i - 1][1]) / 1e6
            if 0 < g < 5.0:          # (step boundaries with host work between them are longer)
                gaps[n][0] += 1
                gaps[n][1] += g
    import os
    if os.environ.get("AIM_PROF_DUMP"):      # raw timeline of the middle of the trace: start us, duration us, gap us, stream, name
        allr = [(a, b, st, short(n)) for st, a, b, n in rows]
        mid = len(allr) // 2
        t0 = allr[mid][0]
        with open(os.environ["AIM_PROF_DUMP"], "w") as f:
            last_end = {}
            for a, b, st, n in allr[mid:mid + 900]:
                g = (a - last_end[st]) / 1e3 if st in last_end else 0.0
                last_end[st] = b
                f.write(f"{(a - t0) / 1e3:10.1f} {(b - a) / 1e3:8.1f} {g:8.1f} s{st} {n[:60]}\n")
    print(f"\n## main stream ({ms}): kernel time and the idle gap before each kernel, ms/step\n")
    print(f"kernels {sum(v[1] for v in agg.values()) / steps:.2f}, gaps {sum(v[1] for v in gaps.values()) / steps:.2f}\n")
    print("| kernel | calls/step | ms/step | avg us | gap before, ms/step | avg gap us |\n|---|---|---|---|---|---|")
    for n, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + gaps[kv[0]][1]))[:40]:
        g = gaps[n]
        print(f"| `{n}` | {v[0] / steps:.1f} | {v[1] / steps:.3f} | {v[1] / v[0] * 1e3:.1f} | {g[1] / steps:.3f} | {g[1] / max(g[0], 1) * 1e3:.1f} |")


if __name__ == "__main__":
    main()
