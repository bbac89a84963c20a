#!/usr/bin/env python
"""Concurrency analysis of a rocprofv3 --kernel-trace database of a pipelined bench run: how many of our kernels are
in flight over time, per-kernel average durations under overlap, per-stream busy fractions and gaps.
usage: rocpd_trace_overlap.py results.db [frames_per_step]"""
import collections
import sqlite3
import statistics
import sys

con = sqlite3.connect(sys.argv[1])
fps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
rows = con.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
fills = [i for i, r in enumerate(rows) if "k_fill_params" in r[0]]
if len(fills) < 8:
    sys.exit("not a pipelined run")
a, bnd = rows[fills[3]][1], rows[fills[-1]][1]
nsteps = len(fills) - 1 - 3
sel = [r for r in rows if a <= r[1] < bnd and "nnn::k_" in r[0]]
wall = (bnd - a) / 1e3
ev = sorted([(s, 1) for _, s, e, _, _ in sel] + [(e, -1) for _, s, e, _, _ in sel])
conc, cur, last = collections.Counter(), 0, ev[0][0]
for t, d in ev:
    conc[cur] += t - last
    last = t
    cur += d
print("wall %.0f us over %d steps of %d frames: %.1f us/frame" % (wall, nsteps, fps, wall / (nsteps * fps)))
print("kernels in flight:", {c: "%.1f%%" % (100 * v / (bnd - a)) for c, v in sorted(conc.items())})
dur = collections.defaultdict(list)
for n, s, e, q, st in sel:
    dur[n.split("(")[0].replace("nnn::", "").replace("void ", "")].append((e - s) / 1e3)
print("avg us (calls per step):", {k: (round(sum(v) / len(v), 1), round(len(v) / nsteps, 1)) for k, v in dur.items()})
print("kernel time per frame: %.1f us" % (sum(sum(v) for v in dur.values()) / (nsteps * fps)))
bys = collections.defaultdict(list)
for n, s, e, q, st in sel:
    bys[(st, q)].append((s, e))
for k, v in bys.items():
    v.sort()
    busy = sum(e - s for s, e in v) / 1e3
    gaps = [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
    print("stream/queue", k, "kernels", len(v), "busy %.0f%%" % (100 * busy / wall), "median gap %.1f us" % statistics.median(gaps))

# idle time in front of each kernel type (a lane waiting for its predecessor group, or for the host)
bk = collections.defaultdict(list)
for n, s, e, q, st in sel:
    bk[(st, q)].append((s, e, n.split("(")[0].replace("nnn::", "").replace("void ", "")))
g = collections.defaultdict(list)
for k, v in bk.items():
    v.sort()
    for i in range(len(v) - 1):
        g[v[i + 1][2]].append((v[i + 1][0] - v[i][1]) / 1e3)
print("idle in front of (mean us, total us per frame):", {k: (round(sum(x) / len(x), 1), round(sum(x) / (nsteps * fps), 1)) for k, x in g.items()})

# textual timeline of ~100 consecutive kernels from the middle of the run (us relative to the first one)
if len(sys.argv) > 3:
    mid = len(sel) // 2
    t0 = sel[mid][1]
    sid = {k: i for i, k in enumerate(sorted(bys))}
    for n, s, e, q, st in sel[mid:mid + int(sys.argv[3])]:
        name = n.split("(")[0].replace("nnn::", "").replace("void ", "")
        print("%8.1f %8.1f  %s%-12s" % ((s - t0) / 1e3, (e - t0) / 1e3, "              " * sid[(st, q)], name))
