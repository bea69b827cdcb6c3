#!/usr/bin/env python3
"""Timeline of one block's backward between two attention-backward launches (rocprofv3 kernel trace): every kernel's
start offset and duration, to see where the class-token chain's time goes.  usage: chain_probe.py <run_results.db>"""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(cur.execute(f"select d.start, d.end, d.queue_id, s.display_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
key = sys.argv[2] if len(sys.argv) > 2 else "attn_bwd_pipe"
idx = [i for i, r in enumerate(rows) if key in r[3] and "cls_" not in r[3]]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = rows[a][0]
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:44]
for s, e, q, n in rows[a:b + 1]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q}  {short(n)}")
