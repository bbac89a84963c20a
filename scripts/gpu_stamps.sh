#!/bin/bash
# Developer tool: build an instrumented library (-DNNN_STAMPS) next to the product one and print the phase breakdown.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
W=$R/nnnoiseless_amd/data/weights.rnn
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -DNNN_STAMPS -I nnnoiseless_amd/csrc -DNNN_WEIGHTS_PATH="\"$W\"" -x hip nnnoiseless_amd/csrc/nnn_batch.hip nnnoiseless_amd/csrc/nnn_model.cpp nnnoiseless_amd/csrc/rnnoise_capi.cpp -o /tmp/libnnn_stamps.so || exit 1
python - <<'PY'
import ctypes as C, numpy as np, sys
sys.path.insert(0, '.')
import nnnoiseless_amd as nn
from nnnoiseless_amd import _ffi
from nnnoiseless_amd.synthetic import make_streams_fast
lib = _ffi.Library('/tmp/libnnn_stamps.so')
lib.L.nnn_batch_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
import os
for S, rows in ((4096, 64), (4096, 16), (65536, 64)):
    os.environ['NNN_RNN_ROWS'] = str(rows)
    bd = nn.BatchDenoiser(S, lib=lib)
    bd.set_graph(False)
    x = make_streams_fast(S, 6)
    bd.process(x)
    st = np.zeros(64, np.int64)
    lib.L.nnn_batch_read_stamps(bd._h, st.ctypes.data_as(C.c_void_p))
    d = lambda a, b: (st[b] - st[a]) / 100.0   # s_memtime ticks at 100 MHz -> us
    print(f"S={S} rnn rows={rows}  [ticks are 100 MHz constant clock -> us]")
    print("  k_hp total", d(24, 25))
    print("  k_lpc: autocorr", d(0, 1), "lpc", d(1, 2), "fir", d(2, 3))
    print("  k_rnn: preload+features+zero", d(8, 9), "split feats", d(9, 10), "dense", d(10, 11), "vad", d(11, 12), "noise(+vadout)", d(12, 13), "dn", d(13, 14), "out", d(14, 15))
    print("  dn layer: stateload->phaseA", d(13, 16), "phaseA", d(16, 17), "rs store", d(17, 18), "phaseB", d(18, 14))
    print("  dn phase A: bias init", d(16, 19), "gemm in", d(19, 20), "gemm zr", d(20, 21), "sigmoids", d(21, 22), "barrier", d(22, 17))
    print("  dn phase B: gemm h", d(18, 23), "epilogue", d(23, 26), "barrier", d(26, 14))
    bd.close()
PY
