import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import nnnoiseless_amd as nn
from oracle import oracle as O
from edge_streams import make_edge_streams
x = make_edge_streams(60)
wb = open('nnnoiseless_amd/data/weights.rnn','rb').read()
ref = O.run_streams(O.Model(wb), x, want=("gains","vad","pitch"))
ref32 = O.run_streams(O.Model(wb, f32_fft=True), x, want=("gains",))
spread = np.abs(ref["gains"] - ref32["gains"]).max(axis=(1, 2))
bd = nn.BatchDenoiser(x.shape[0])
worst = np.zeros(x.shape[0])
for t in range(x.shape[1]):
    bd.process(x[:, t:t+1])
    g = bd.tap("g")
    worst = np.maximum(worst, np.abs(g - ref["gains"][:, t]).max(axis=1))
print(os.environ.get("NNN_LIBRARY","default").split("/")[-1], "gerr/spread", np.round(worst / np.maximum(spread, 1e-4/3), 2), "spread", np.round(spread*1e4,2))
