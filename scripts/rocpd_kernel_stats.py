#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace as a per-kernel stats table (markdown / csv).
usage: rocpd_kernel_stats.py trace_results.db [out.md]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = con.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                   f"from kernels group by {name_col} order by sum(end - start) desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, c, tot, avg, mn, mx in rows:
    short = n.split("(")[0].replace("nnn::", "")
    lines.append(f"| {short} | {c} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")
text = "\n".join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
