"""Latency of the drop-in single-stream surface (a batch of one): DenoiseState.process_frame and a 256-stream host call."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import nnnoiseless_amd as nn
st = nn.DenoiseState.new()
x = (1000 * np.sin(2 * np.pi * 440 * np.arange(480 * 300) / 48000)).astype(np.float32).reshape(300, 480)
out = np.zeros(480, np.float32)
for f in x[:50]:
    st.process_frame(out, f)
t0 = time.perf_counter()
for f in x[50:]:
    st.process_frame(out, f)
dt = time.perf_counter() - t0
print("DenoiseState.process_frame (1 stream, host buffers): %.0f us per frame = %.0f frames/s" % (dt / 250 * 1e6, 250 / dt))
bd = nn.BatchDenoiser(256)
xb = np.tile(x[None, :100], (256, 1, 1))
bd.process(xb[:, :10])
t0 = time.perf_counter()
for t in range(10, 100):
    bd.process(xb[:, t:t + 1])
dt = time.perf_counter() - t0
print("BatchDenoiser(256).process, 1 frame per call, host buffers: %.0f us per call = %.0f frames/s" % (dt / 90 * 1e6, 256 * 90 / dt))
