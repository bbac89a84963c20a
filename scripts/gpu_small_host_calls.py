import sys, time, os
import numpy as np
sys.path.insert(0, ".")
import nnnoiseless_amd as nn
x = (1000 * np.sin(2 * np.pi * 440 * np.arange(480 * 120) / 48000)).astype(np.float32).reshape(120, 480)
for S in [int(a) for a in sys.argv[1:]] or (1, 8, 32, 64, 128, 136):
    bd = nn.BatchDenoiser(S, max_group_frames=1)
    xb = np.ascontiguousarray(np.tile(x[None], (S, 1, 1)))
    for t in range(20): bd.process(xb[:, t:t + 1])
    t0 = time.perf_counter()
    for t in range(20, 120): bd.process(xb[:, t:t + 1])
    dt = time.perf_counter() - t0
    print(f"S={S}: {dt / 100 * 1e6:.0f} us per one-frame host call", flush=True)
    bd.close()
