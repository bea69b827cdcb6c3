#!/usr/bin/env python3
"""Per-launch PMC summary of the large GEMM kernels from three separate rocprofv3 passes (FETCH_SIZE | WRITE_SIZE | SQ set).

usage: pmc_summary.py <fetch.db> <write.db> <sq.db> > profiles/rNN_gemm_pmc.json
* FETCH_SIZE / WRITE_SIZE: KB units (x1024); FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM: 128-B requests are
  tallied at 64 B).  These are L2 -> fabric requests: Infinity-Cache hits are counted, so this is fabric traffic, an upper
  bound on HBM traffic.
* MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 4 SIMDs * 256 CUs) -- the derived-metric formula of
  rocprofv3's MfmaUtil with the GUI-active cycles summed over the 8 XCDs divided by 8.
"""
import collections
import json
import re
import sqlite3
import sys


def per_kernel(db, counters):
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    ev = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.display_name, i.name, e.value, d.end - d.start, d.event_id from {ev} e join {info} i on e.pmc_id = i.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for name, cname, val, dur, evid in cur.execute(q):
        m = re.search(r"gemm256_kernel<(\d), (false|true)>", name)
        if not m or cname not in counters:
            continue
        k = "gemm256_kernel<%s%s>" % (m.group(1), ", fp8" if m.group(2) == "true" else "")
        agg[k][cname] += float(val)
        if evid not in seen[k]:
            seen[k].add(evid)
            agg[k]["_launches"] += 1
            agg[k]["_ns"] += dur
    return agg


def main():
    f = per_kernel(sys.argv[1], {"FETCH_SIZE"})
    w = per_kernel(sys.argv[2], {"WRITE_SIZE"})
    sq = per_kernel(sys.argv[3], {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES"})
    out = {"source": "rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing "
                     "--no-secondary --no-inference; three separate passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
                     "SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES)",
           "names": "gemm256_kernel<E>: E = 0 EPI_BF16, 1 EPI_ACT, 2 EPI_DACT, 3 EPI_F32, 4 EPI_EXPSUM",
           "kernels": {}}
    for k in sorted(f):
        n = f[k]["_launches"]
        fb = f[k]["FETCH_SIZE"] / n * 1024 * 2
        wb = w[k]["WRITE_SIZE"] / max(w[k]["_launches"], 1) * 1024
        d = {"launches": int(n), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
        if k in sq and sq[k]["GRBM_GUI_ACTIVE"] > 0:
            s = sq[k]
            gui = s["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
            d["mfma_busy_frac"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 4 * 256)
            d["avg_us_in_pmc_pass"] = s["_ns"] / s["_launches"] / 1e3
            d["eff_clock_ghz"] = gui / s["_ns"]
            d["mfma_mops_bf16_per_launch"] = s["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / s["_launches"]
            d["flops_from_mops"] = d["mfma_mops_bf16_per_launch"] * 512
        out["kernels"][k] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
