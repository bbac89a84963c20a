"""Regenerates pitch_kat.json from the CPU oracle (derived known answers, not reference-issued)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402
from nnnoiseless_amd.synthetic import make_streams  # noqa: E402

model = O.Model(open(os.path.join(HERE, "..", "..", "nnnoiseless_amd", "data", "weights.rnn"), "rb").read())
inp = np.fromfile(os.path.join(HERE, "testing.raw"), dtype="<i2").astype(np.float32)
x = inp[: 100 * 480].reshape(1, 100, 480)
r = O.run_streams(model, x)
syn = O.run_streams(model, make_streams(0, 16, 40))
json.dump({"testing_raw": r["pitch"][0].tolist(), "synthetic_s0_16_f40": syn["pitch"].tolist()},
          open(os.path.join(HERE, "pitch_kat.json"), "w"))
