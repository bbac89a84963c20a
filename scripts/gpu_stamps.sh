#!/bin/bash
# Developer tool: build an instrumented library (-DNNN_STAMPS) next to the product one and print k_rnn's phase breakdown
# (shader-clock stamps of block 0 / thread 0; the frame loop overwrites them, so they describe the LAST frame of the launch).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
W=$R/nnnoiseless_amd/data/weights.rnn
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-unused-value -DNNN_STAMPS -I nnnoiseless_amd/csrc -DNNN_WEIGHTS_PATH="\"$W\"" -x hip nnnoiseless_amd/csrc/nnn_batch.hip nnnoiseless_amd/csrc/nnn_resample.hip nnnoiseless_amd/csrc/nnn_model.cpp nnnoiseless_amd/csrc/rnnoise_capi.cpp nnnoiseless_amd/csrc/nnn_node.cpp -o /tmp/libnnn_stamps.so || exit 1
python - <<'PY'
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, '.')
import nnnoiseless_amd as nn
from nnnoiseless_amd import _ffi
from nnnoiseless_amd.synthetic import make_streams_fast
lib = _ffi.Library('/tmp/libnnn_stamps.so')
lib.L.nnn_batch_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
for S, rows, T in ((4096, 0, 4), (4096, 32, 4), (65536, 0, 4)):
    os.environ.pop('NNN_RNN_ROWS', None)
    if rows: os.environ['NNN_RNN_ROWS'] = str(rows)
    bd = nn.BatchDenoiser(S, lib=lib)
    bd.set_pipeline(False)
    x = make_streams_fast(S, 2 * T)
    bd.process(x[:, :T]); bd.process(x[:, T:])
    st = np.zeros(64, np.int64)
    lib.L.nnn_batch_read_stamps(bd._h, st.ctypes.data_as(C.c_void_p))
    d = lambda a, b: round((st[b] - st[a]) / 2100.0, 2)   # shader-clock cycles at ~2.1 GHz -> us (approximate)
    print(f"S={S} rnn rows={rows} frames/launch={T}  [us]")
    print("  k_rnn: setup+feat0", d(8, 9), "| last frame: copy..dense", d(10, 11), "vad", d(11, 12), "noise(+vadout)", d(12, 13), "dn", d(13, 14), "out+featnext", d(14, 15), " frame total", d(10, 15))
    print("  dn layer: frag issue..phaseA done", d(16, 22), "barrier wait", d(22, 18), "gemm h", d(18, 23), "epilogue", d(23, 26), "closing barrier", d(26, 14))
    print("  k_rnn_wf: setup", d(50, 51), "ticks", d(51, 52), "| tick 2 by role [first phase, barrier wait, second phase, barrier wait]:")
    for r, name in enumerate(("denoise unit", "noise unit", "vad + out + dense", "features")):
        o = 30 + 5 * r
        print("     ", name, d(o, o + 1), d(o + 1, o + 2), d(o + 2, o + 3), d(o + 3, o + 4), " tick", d(o, o + 4))
    print("  k_hp total (last frame)", d(24, 25), " (k_pitch: scripts/gpu_stamps_pitch.sh)")
    bd.close()
PY
