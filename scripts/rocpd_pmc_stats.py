#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd (sqlite) database.
usage: rocpd_pmc_stats.py pmc_results.db"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute(
    "select k.name, e.counter_name, count(*), avg(e.counter_value), sum(e.counter_value) "
    "from pmc_events e join kernels k on k.dispatch_id = e.dispatch_id "
    "group by k.name, e.counter_name order by sum(e.counter_value) desc").fetchall()
out = {}
for name, cname, n, avg, tot in rows:
    short = name.split("(")[0].replace("nnn::", "")
    print(f"{short:40s} {cname:12s} launches={n:4d} avg={avg:14.1f} total={tot:14.1f}")
