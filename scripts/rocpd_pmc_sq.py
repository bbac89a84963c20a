#!/usr/bin/env python
"""Per-kernel SQ counters of two rocprofv3 --pmc passes (rocpd sqlite) as JSON, with the busy fractions bench.py quotes.
usage: rocpd_pmc_sq.py streams a_results.db [b_results.db ...] > profiles/pmc_sq_<streams>streams.json

Counters are averages per launch of per-shader-engine samples (32 SEs of 8 CUs = 32 SIMDs each on MI355X); SQ_ACTIVE_INST_* and
SQ_WAVE_CYCLES count quad-cycles, SQ_BUSY_CYCLES and SQ_LDS_IDX_ACTIVE cycles (MI355X_MICROARCH.md, per-instruction constants).
  valu_busy = 4 * SQ_ACTIVE_INST_VALU / (32 SIMDs * SQ_BUSY_CYCLES)     share of SIMD issue slots taken by vector-ALU work
  lds_busy  = SQ_LDS_IDX_ACTIVE / (8 CUs * SQ_BUSY_CYCLES)                share of LDS-array cycles in use
  lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"""
import collections
import json
import sqlite3
import sys

S = int(sys.argv[1])
t = collections.defaultdict(dict)
for db in sys.argv[2:]:
    con = sqlite3.connect(db)
    rows = con.execute("select k.name, e.counter_name, avg(e.counter_value) from pmc_events e join kernels k "
                       "on k.dispatch_id = e.dispatch_id group by k.name, e.counter_name").fetchall()
    for n, c, avg in rows:
        t[n.split("(")[0].replace("nnn::", "").replace("void ", "").split("<")[0]][c] = avg
out = {"streams": S, "frames_per_launch": 24, "kernels": {}, "note": __doc__.split("\n\n", 1)[1]}
for k, v in sorted(t.items()):
    if not k.startswith("k_") or k == "k_fill_params":
        continue
    name = "k_rnn" if k == "k_rnn_wf" else ("k_hp" if k == "k_hp2" else k)
    busy = v.get("SQ_BUSY_CYCLES", 0.0)
    e = {"counters": {c: v[c] for c in sorted(v)}, "kernel_symbol": k}
    if busy:
        e["valu_busy"] = 4.0 * v.get("SQ_ACTIVE_INST_VALU", 0.0) / (32.0 * busy)
        e["lds_busy"] = v.get("SQ_LDS_IDX_ACTIVE", 0.0) / (8.0 * busy)
    if v.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
    if name in out["kernels"] and out["kernels"][name]["counters"].get("SQ_WAVES", 0) > v.get("SQ_WAVES", 0):
        continue   # (k_rnn and k_rnn_wf both present: keep the one that did the work)
    out["kernels"][name] = e
print(json.dumps(out, indent=1))
