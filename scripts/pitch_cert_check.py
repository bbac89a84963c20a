"""The certified coarse pitch search (k_pitch, round 6) against the oracle, frame by frame, through the taps of mode 2: the lags it kept
carry the exact sums, the pair it returns is the oracle's, and how many lags survive.   usage: pitch_cert_check.py [streams] [frames] [hostsim]"""
import os
import sys

import numpy as np

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
import nnnoiseless_amd as nn  # noqa: E402
from nnnoiseless_amd import _ffi  # noqa: E402
from nnnoiseless_amd.synthetic import make_streams  # noqa: E402
from oracle import oracle  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
lib = None
if len(sys.argv) > 3 and sys.argv[3] == "hostsim":
    sys.path.insert(0, os.path.join(R, "tests", "hostsim"))
    import build_hostsim
    lib = _ffi.Library(build_hostsim.build())
x = make_streams(0, S, T)
bd = nn.BatchDenoiser(S, lib=lib)
bd.set_taps(2)
model = oracle.Model(open(os.path.join(R, "nnnoiseless_amd", "data", "weights.rnn"), "rb").read())
sts = [oracle.State(model) for _ in range(S)]
bad, nsv = 0, []
for t in range(T):
    bd.process(x[:, t:t + 1])
    xc, b1, ps, pi = bd.tap("xcorr1"), bd.tap("best1"), bd.tap("pitch_search"), bd.tap("pitch")
    for s in range(S):
        sts[s].process_frame(x[s, t])
        tp = sts[s].taps()
        keep = ~np.isnan(xc[s])
        nsv.append(int(keep.sum()))
        if not np.array_equal(xc[s][keep], tp["xcorr1"][keep]) or not np.array_equal(b1[s], tp["best1"]) or ps[s, 0] != tp["pitch_search"] or pi[s, 0] != tp["pitch_idx"]:
            bad += 1
            print("MISMATCH frame", t, "stream", s, b1[s], tp["best1"], int(keep.sum()))
nsv = np.array(nsv)
print(f"frames {len(nsv)}  mismatches {bad}  survivors per stream-frame: mean {nsv.mean():.2f}  max {nsv.max()}  all 147 (full search): {(nsv == 147).mean() * 100:.2f} %")
print("histogram 0..24:", np.bincount(np.minimum(nsv, 25))[:26])
sys.exit(1 if bad else 0)
