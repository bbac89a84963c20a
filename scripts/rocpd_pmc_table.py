#!/usr/bin/env python
"""Per-kernel table of every PMC counter in one or more rocprofv3 rocpd (sqlite) databases.
usage: rocpd_pmc_table.py a_results.db [b_results.db ...]   (averages per launch; SQ counters are per-SE samples)"""
import collections
import sqlite3
import sys

t = collections.defaultdict(dict)
for db in sys.argv[1:]:
    con = sqlite3.connect(db)
    rows = con.execute("select k.name, e.counter_name, avg(e.counter_value) from pmc_events e join kernels k "
                       "on k.dispatch_id = e.dispatch_id group by k.name, e.counter_name").fetchall()
    for n, c, avg in rows:
        t[n.split("(")[0].replace("nnn::", "").replace("void ", "")][c] = avg
cs = sorted({c for v in t.values() for c in v})
print("| kernel | " + " | ".join(c.replace("SQ_", "") for c in cs) + " |")
print("|---|" + "---|" * len(cs))
for k, v in sorted(t.items()):
    if k.startswith("k_"):
        print(f"| {k} | " + " | ".join("%.4g" % v.get(c, 0) for c in cs) + " |")
