"""Synthetic 48 kHz mono streams: sine + noise in i16 range, delivered as f32.

Per stream s the parameters come from numpy.random.default_rng(0xD3701 + s):
f ~ U[80,1000] Hz, A ~ U[500,12000], sigma ~ U[50,3000], phi ~ U[0,2pi).  One stream in 16 is pure
silence (exercises the e < 0.04 gate), one in 16 is noise only (exercises corr <= 0).
"""
import numpy as np

SEED0 = 0xD3701


def stream_params(s):
    rng = np.random.default_rng(SEED0 + s)
    f = rng.uniform(80.0, 1000.0)
    a = rng.uniform(500.0, 12000.0)
    sigma = rng.uniform(50.0, 3000.0)
    phi = rng.uniform(0.0, 2.0 * np.pi)
    kind = "silence" if s % 16 == 7 else ("noise" if s % 16 == 11 else "tone")
    return f, a, sigma, phi, kind, rng


def make_streams(first, count, n_frames):
    """[count, n_frames, 480] float32; stream `first + i` is reproducible independently of the batch."""
    n = np.arange(n_frames * 480, dtype=np.float64)
    x = np.zeros((count, n_frames * 480), np.float32)
    for i in range(count):
        f, a, sigma, phi, kind, rng = stream_params(first + i)
        if kind == "silence":
            continue
        v = sigma * rng.standard_normal(n.size)
        if kind == "tone":
            v = v + a * np.sin(2.0 * np.pi * f * n / 48000.0 + phi)
        x[i] = np.clip(np.round(v), -32768, 32767)
    return x.reshape(count, n_frames, 480)


def make_streams_fast(count, n_frames, seed=0):
    """Same distribution, one vectorised generator for the whole batch (bench-sized inputs)."""
    rng = np.random.default_rng(SEED0 ^ (seed + 1))
    n = np.arange(n_frames * 480, dtype=np.float32)[None, :]
    f = rng.uniform(80.0, 1000.0, (count, 1)).astype(np.float32)
    a = rng.uniform(500.0, 12000.0, (count, 1)).astype(np.float32)
    sigma = rng.uniform(50.0, 3000.0, (count, 1)).astype(np.float32)
    phi = rng.uniform(0.0, 2.0 * np.pi, (count, 1)).astype(np.float32)
    idx = np.arange(count) % 16
    a[idx == 11] = 0.0
    x = a * np.sin((2.0 * np.pi / 48000.0) * f * n + phi)
    x += sigma * rng.standard_normal(x.shape, dtype=np.float32)
    x[idx == 7] = 0.0
    return np.clip(np.round(x), -32768, 32767).astype(np.float32).reshape(count, n_frames, 480)


def make_streams_device(torch, dev, count, n_frames, seed=0):
    """Same distribution as make_streams_fast, generated on `dev` with torch (bench-sized inputs: 65 536 streams x tens of
    frames would take minutes in numpy).  Returns a float32 tensor [count, n_frames, 480] resident on the device."""
    g = torch.Generator(device=dev)
    g.manual_seed(SEED0 ^ (seed + 1))
    u = torch.rand((4, count, 1), generator=g, device=dev, dtype=torch.float32)
    f = 80.0 + 920.0 * u[0]
    a = 500.0 + 11500.0 * u[1]
    sigma = 50.0 + 2950.0 * u[2]
    phi = 2.0 * np.pi * u[3]
    idx = torch.arange(count, device=dev) % 16
    a[idx == 11] = 0.0
    n = n_frames * 480
    x = torch.empty((count, n), dtype=torch.float32, device=dev)
    chunk = max(1, int(2 ** 27 // max(n, 1)))           # bound the temporaries (~0.5 GB each)
    t = torch.arange(n, device=dev, dtype=torch.float32)[None, :]
    for lo in range(0, count, chunk):
        hi = min(count, lo + chunk)
        v = a[lo:hi] * torch.sin((2.0 * np.pi / 48000.0) * f[lo:hi] * t + phi[lo:hi])
        v += sigma[lo:hi] * torch.randn((hi - lo, n), generator=g, device=dev, dtype=torch.float32)
        x[lo:hi] = torch.clamp(torch.round(v), -32768.0, 32767.0)
    x[idx == 7] = 0.0
    return x.reshape(count, n_frames, 480)
