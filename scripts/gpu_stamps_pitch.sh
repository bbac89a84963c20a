#!/bin/bash
# Developer tool: phase breakdown of k_pitch as block 0 / wave 0 sees it (shader-clock stamps of the LAST frame of a launch; the
# times include that wave's waits at the block barriers).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
W=$R/nnnoiseless_amd/data/weights.rnn
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-unused-value -DNNN_STAMPS ${STAMP_FLAGS:-} -I nnnoiseless_amd/csrc -DNNN_WEIGHTS_PATH="\"$W\"" -x hip nnnoiseless_amd/csrc/nnn_batch.hip nnnoiseless_amd/csrc/nnn_resample.hip nnnoiseless_amd/csrc/nnn_model.cpp nnnoiseless_amd/csrc/rnnoise_capi.cpp nnnoiseless_amd/csrc/nnn_node.cpp -o /tmp/libnnn_stamps.so || exit 1
python - <<'PY'
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, '.')
import nnnoiseless_amd as nn
from nnnoiseless_amd import _ffi
from nnnoiseless_amd.synthetic import make_streams_fast
lib = _ffi.Library('/tmp/libnnn_stamps.so')
lib.L.nnn_batch_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
for S, T in ((256, 2), (4096, 4)):   # (short launches whose frames run side by side with the flag hand-off: every stamp of block 0 is one frame's; in a launch that
                                      # loops over 24 frames the slots written by other waves belong to other frames)
    bd = nn.BatchDenoiser(S, lib=lib)
    bd.set_pipeline(False)
    x = make_streams_fast(S, 2 * T)
    bd.process(x[:, :T]); bd.process(x[:, T:])
    st = np.zeros(64, np.int64)
    lib.L.nnn_batch_read_stamps(bd._h, st.ctypes.data_as(C.c_void_p))
    names = ["window->lds", "fir", "matrix products, survivors | scan starts", "full search", "exact sums | scans", "find_best", "fine xcorr", "combine+energies", "replay", "candidates", "cand inner", "judge k", "pick", "refine", "final"]
    idx = [0, 1, 4, 5, 63, 27, 6, 7, 59, 60, 53, 54, 58, 55, 56, 57]
    d = [(st[idx[i + 1]] - st[idx[i]]) / 2100.0 for i in range(len(names))]   # shader-clock cycles at ~2.1 GHz -> us (approximate)
    print(f"S={S} k_pitch last frame of block 0 [us, approximate]: total {sum(d):.1f}")
    print("   " + "  ".join(f"{n} {v:.2f}" for n, v in zip(names, d)))
    u = lambda a, b_: (st[a] - st[b_]) / 2100.0
    print(f"   roles, us after the FIR's barrier: matrix products done {u(2, 4):.2f}, fine-lag start sum {u(28, 4):.2f}, yy start sum {u(29, 4):.2f}, coarse-lag energies' first piece {u(3, 4):.2f}")
    bd.close()
PY
