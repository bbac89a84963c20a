#!/bin/bash
# Developer tool: stretch-by-stretch breakdown of k_back (fused, and its RNN stretch alone) as block 0 / wave 0 sees it (shader-clock
# stamps), one-frame launches and 24-frame groups; then a rocprofv3 kernel trace of one-frame calls (true kernel durations and gaps).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
W=$R/nnnoiseless_amd/data/weights.rnn
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-unused-value -DNNN_STAMPS ${NNN_EXTRA:-} -I nnnoiseless_amd/csrc -DNNN_WEIGHTS_PATH="\"$W\"" -x hip nnnoiseless_amd/csrc/nnn_batch.hip nnnoiseless_amd/csrc/nnn_resample.hip nnnoiseless_amd/csrc/nnn_model.cpp nnnoiseless_amd/csrc/rnnoise_capi.cpp nnnoiseless_amd/csrc/nnn_node.cpp -o /tmp/libnnn_stamps.so || exit 1
python - <<'PY' 2>&1 | grep -v Warning | tee gpurun_out/r4_back_stamps${TAG:-}.txt
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, '.')
import nnnoiseless_amd as nn
from nnnoiseless_amd import _ffi
from nnnoiseless_amd.synthetic import make_streams_fast
lib = _ffi.Library('/tmp/libnnn_stamps.so')
lib.L.nnn_batch_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
names = ["prologue", "(loop)", "transforms+head", "features", "barrier", "IN fill", "dense", "vad GRU", "noise GRU", "denoise GRU", "output+barrier", "synthesis", "epilogue"]
for S in (4096,):
    x = make_streams_fast(S, 30)
    for mode, T in ((1, 1), (3, 1), (2, 24), (4, 24)):
        bd = nn.BatchDenoiser(S, lib=lib)
        bd.set_back_end(mode)
        bd.set_pipeline(False)
        bd.process(x[:, :3]); 
        for t in range(3, 6): bd.process(x[:, t:t + 1])
        bd.process(x[:, 6:6 + T])
        st = np.zeros(64, np.int64)
        lib.L.nnn_batch_read_stamps(bd._h, st.ctypes.data_as(C.c_void_p))
        us = lambda a, b: (st[b] - st[a]) / 2100.0
        parts = "  ".join(f"{n} {us(i, i + 1):.2f}" for i, n in enumerate(names) if (mode in (1, 2) or i not in (11,)) and i != 1)
        print(f"S={S} mode {mode} frames/launch {T}: whole kernel {us(0, 13):.1f} us; last frame {us(2, 12 if mode in (1, 2) else 11):.2f} us | {parts}")
        bd.close()
PY
[ -n "${NO_TRACE:-}" ] && exit 0
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r4_tick_trace
for M in 0 1 3; do
  NNN_BACK=$M timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4_tick_trace/mode$M -- python $R/scripts/tick_capacity.py 4096 1 300 1 1 2>&1 | grep "M frames"
  DB=$(find $R/gpurun_out/r4_tick_trace/mode$M -name '*_results.db' | head -1)
  python $R/scripts/rocpd_kernel_stats.py "$DB" | head -12
  python - "$DB" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select start, end from kernels order by start").fetchall()
rows = rows[len(rows) // 2:]                     # the steady state
gaps = [b[0] - a[1] for a, b in zip(rows, rows[1:])]
busy = sum(e - s for s, e in rows)
span = rows[-1][1] - rows[0][0]
print(f"steady state: {len(rows)} launches, kernels busy {busy / span:.3f} of the time, mean gap between launches {sum(gaps) / len(gaps) / 1e3:.2f} us")
PY
done 2>&1 | tee $R/gpurun_out/r4_tick_trace.txt
find $R/gpurun_out/r4_tick_trace -name '*.db' -delete
