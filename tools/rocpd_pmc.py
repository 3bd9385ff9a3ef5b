#!/usr/bin/env python3
"""Per-kernel PMC totals from a rocprofv3 (rocpd sqlite) counter-collection run.
usage: rocpd_pmc.py results.db -> markdown table (kernel, dispatches, mean counter value per dispatch)"""
import re
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
print("<!-- columns:", cols, "-->")
name_col = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x and "counter" not in x][0]
cn = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x and "name" in x][0]
cv = "value" if "value" in cols else [x for x in cols if "value" in x][0]
did = "dispatch_id" if "dispatch_id" in cols else None
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
q = f"select {name_col}, {cn}, {cv}" + (f", {did}" if did else "") + " from counters_collection"
seen = defaultdict(float)
for row in c.execute(q):
    k = re.sub(r"\(.*", "", row[0]).replace("void ", "")
    if did:
        seen[(k, row[1], row[3])] += row[2]
    else:
        a = agg[k][row[1]]
        a[0] += 1
        a[1] += row[2]
if did:
    for (k, cname, _), v in seen.items():
        a = agg[k][cname]
        a[0] += 1
        a[1] += v
print("| kernel | counter | dispatches | mean per dispatch |")
print("|---|---|---:|---:|")
for k, d in sorted(agg.items()):
    for cname, (n, tot) in sorted(d.items()):
        print(f"| {k} | {cname} | {n} | {tot / n:.1f} |")
