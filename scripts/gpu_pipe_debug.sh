#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python -X faulthandler - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
import nnnoiseless_amd as nn
from nnnoiseless_amd.synthetic import make_streams
x = make_streams(0, 64, 20)
ref = nn.BatchDenoiser(64); ref.set_pipeline(False)
want, wv = ref.process(x)
print("reference (single-frame graphs) ok", flush=True)
bd = nn.BatchDenoiser(64)
got, gv = bd.process(x)
print("pipelined ok; identical:", np.array_equal(got, want), np.array_equal(gv, wv), flush=True)
PY
