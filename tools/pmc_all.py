#!/usr/bin/env python3
"""Per-kernel fabric traffic of one bench step from two rocprofv3 PMC passes (FETCH_SIZE | WRITE_SIZE), every kernel.

usage: pmc_all.py <fetch.db> <write.db> [steps] > profiles/rNN_traffic_by_kernel.md
FETCH_SIZE / WRITE_SIZE are KB (x1024); FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md: 128-B requests tallied at
64 B).  These are L2 -> fabric requests (Infinity-Cache hits included): an upper bound on HBM traffic.  Kernels are
serialised by the profiler in a PMC pass, so durations are stand-alone times, not in-step times."""
import collections
import re
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    ev = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.display_name, e.value, d.end - d.start from {ev} e join {info} i on e.pmc_id = i.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id where i.name = ?")
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for name, val, dur in cur.execute(q, (counter,)):
        k = name.replace("void ", "").replace("(anonymous namespace)::", "")
        k = re.sub(r"\(.*", "", k)[:60]
        a = agg[k]
        a[0] += 1; a[1] += float(val); a[2] += dur
    return agg


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
    rows = []
    for k in f:
        n = f[k][0]
        fb = f[k][1] * 1024 * 2 / steps
        wb = w[k][1] * 1024 / steps if k in w else 0.0
        rows.append((fb + wb, k, n / steps, fb, wb, f[k][2] / steps / 1e6))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("# fabric traffic per step by kernel (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, separate passes, kernels serialised)\n")
    print(f"total {tot / 1e9:.1f} GB per step; stand-alone kernel time {sum(r[5] for r in rows):.1f} ms per step\n")
    print("| kernel | launches/step | fetch GB | write GB | total GB | ms/step stand-alone | TB/s |")
    print("|---|---|---|---|---|---|---|")
    for t, k, n, fb, wb, ms in rows:
        if t < 1e6:
            continue
        print(f"| `{k}` | {n:.1f} | {fb / 1e9:.2f} | {wb / 1e9:.2f} | {t / 1e9:.2f} | {ms:.2f} | {t / max(ms, 1e-9) / 1e9:.2f} |")


if __name__ == "__main__":
    main()
