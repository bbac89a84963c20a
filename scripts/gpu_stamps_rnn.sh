#!/bin/bash
# Developer tool: role-by-role phase breakdown of one mid-group tick of k_rnn_wf as block 0 sees it (shader-clock stamps).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
W=$R/nnnoiseless_amd/data/weights.rnn
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-unused-value -DNNN_STAMPS -DNNN_WFSTAMP_PHASE=${WFPHASE:-1} ${NNN_EXTRA:-} -I nnnoiseless_amd/csrc -DNNN_WEIGHTS_PATH="\"$W\"" -x hip nnnoiseless_amd/csrc/nnn_batch.hip nnnoiseless_amd/csrc/nnn_resample.hip nnnoiseless_amd/csrc/nnn_model.cpp nnnoiseless_amd/csrc/rnnoise_capi.cpp nnnoiseless_amd/csrc/nnn_node.cpp -o /tmp/libnnn_stamps.so || exit 1
python - <<'PY'
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, '.')
import nnnoiseless_amd as nn
from nnnoiseless_amd import _ffi
from nnnoiseless_amd.synthetic import make_streams_fast
lib = _ffi.Library('/tmp/libnnn_stamps.so')
lib.L.nnn_batch_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
for S, T in ((4096, 16), (65536, 16)):
    bd = nn.BatchDenoiser(S, lib=lib)
    bd.set_pipeline(False)
    x = make_streams_fast(S, 2 * T)
    bd.process(x[:, :T]); bd.process(x[:, T:])
    st = np.zeros(64, np.int64)
    lib.L.nnn_batch_read_stamps(bd._h, st.ctypes.data_as(C.c_void_p))
    us = lambda a, b: (st[b] - st[a]) / 2100.0
    print(f"S={S} k_rnn_wf block 0: prologue {us(50, 51):.2f} us, {T} frames {us(51, 52):.2f} us ({us(51, 52) / (T + 4):.2f} per tick)")
    t0 = min(st[30 + 5 * r] for r in range(4))
    for r, name in enumerate(("denoise wave 0", "noise wave 6", "vad/dense wave 9", "features wave 11")):
        o = 30 + 5 * r
        print(f"   {name:18s} start +{(st[o] - t0) / 2100.0:.2f}  phase 1 {us(o, o + 1):.2f}  wait {us(o + 1, o + 2):.2f}  phase 2 {us(o + 2, o + 3):.2f}  wait {us(o + 3, o + 4):.2f}")
    print("   features wave, frame 3: " + "  ".join(f"{n} {us(8 + i, 9 + i):.2f}" for i, n in enumerate(("stage cepstrum", "ring update", "7 distances", "sync + 40 outputs", "pitch + variability"))))
    ph = int(os.environ.get("WFPHASE", "1"))
    ref = st[30] if ph == 1 else st[32]
    print(f"   every wave, end of phase {ph} after its start [us]: " + " ".join(f"{(st[14 + w] - ref) / 2100.0:.2f}" for w in range(12)))
    bd.close()
PY
