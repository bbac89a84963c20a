#!/usr/bin/env python
"""HBM-side bytes per launch of every kernel from two rocprofv3 --pmc passes (rocpd sqlite databases).
usage: rocpd_pmc_traffic.py FETCH_results.db WRITE_results.db STREAMS FRAMES_PER_LAUNCH > profiles/pmc_traffic_<S>streams.json

FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE reports half of coalesced streaming reads
(MI355X_MICROARCH.md, HBM section), so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; the factor is re-checked
in the same run on torch's elementwise abs kernel (reads N bytes, writes N bytes)."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select k.name, avg(e.counter_value), count(*) from pmc_events e join kernels k on k.dispatch_id = e.dispatch_id "
                       "where e.counter_name = ? group by k.name", (counter,)).fetchall()
    return {n: (v, c) for n, v, c in rows}


fetch, write, streams = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
fpl = int(sys.argv[4]) if len(sys.argv) > 4 else 1
out = {"streams": streams, "frames_per_launch": fpl, "kernels": {}, "calibration": {},
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (scripts/gpu_pmc_traffic.sh, bench.py "
               "--frames-per-step 24: every launch covers a full group of 24 frames of every stream, the production group length); KB per launch; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reports half "
               "of coalesced streaming reads, MI355X_MICROARCH.md HBM section; see `calibration`: torch's abs kernel over the "
               "bench input reads and writes the same number of bytes).  At 4096 streams the working set sits in the 256 MiB "
               "Infinity Cache, whose hits these fabric-side counters include."}
for name in sorted(set(fetch) | set(write)):
    f, w = fetch.get(name, (0.0, 0))[0], write.get(name, (0.0, 0))[0]
    short = name.split("(")[0].replace("nnn::", "").replace("void ", "")
    entry = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024,
             "hbm_bytes_per_stream_frame": (2 * f + w) * 1024 / (streams * fpl)}
    if short.startswith("k_"):
        if "k_fill_params" in short:
            continue
        out["kernels"][short.split("<")[0].replace("k_rnn_wf", "k_rnn").replace("k_hp2", "k_hp")] = entry   # (k_hp2: the two-wave high-pass of small launches)
    elif "AbsFunctor" in name:
        out["calibration"]["torch_abs"] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "fetch_over_write": f / w if w else None}
print(json.dumps(out, indent=1))
