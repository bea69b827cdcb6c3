#!/usr/bin/env python3
"""HBM-side bytes per launch of the large GEMM kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_traffic.py <fetch run_results.db> <write run_results.db> > profiles/rNN_gemm_traffic.json
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of the counter definition (value x 1024 bytes here: the derived
metric is in KB); FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM: 128-B requests tallied at 64 B)."""
import collections
import json
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
    q = (f"select s.display_name, e.value from {ev} e join {info} i on e.pmc_id = i.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id where i.name = ?")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, val in cur.execute(q, (counter,)):
        m = re.search(r"gemm256_kernel<(\d)>", name)
        if m:
            agg["gemm256_kernel<%s>" % m.group(1)][0] += 1
            agg["gemm256_kernel<%s>" % m.group(1)][1] += float(val)
    return agg


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 bench.py "
                     "--steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing; counter values are KB (x1024); FETCH_SIZE doubled "
                     "per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)",
           "kernels": {}}
    for k in sorted(f):
        n = f[k][0]
        fb = f[k][1] / n * 1024 * 2
        wb = w[k][1] / max(w[k][0], 1) * 1024
        out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
