import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hostsim")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # the host's setting for batches driven side by side: the evidence stays comparable
GOLDEN = os.path.join(ROOT, "tests", "golden")
WEIGHTS = os.path.join(ROOT, "nnnoiseless_amd", "data", "weights.rnn")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_metric(out_f32, ref_i16):
    """The reference's own acceptance metric, src/lib.rs:184-194: outputs cast `as i16`
    (truncate toward zero, saturating), sum (ref-out)^2 / sum out^2."""
    o16 = np.clip(np.trunc(out_f32), -32768, 32767).astype(np.int16)
    xx = float((o16.astype(np.float64) ** 2).sum())
    diff = float(((ref_i16.astype(np.float64) - o16.astype(np.float64)) ** 2).sum())
    return diff / xx


@pytest.fixture(scope="session", autouse=True)
def _torch_before_the_library():
    """On a GPU box, initialise torch's HIP runtime before the library's first call: torch ships its own copy of the runtime, and a
    process that touches the device through the system copy first finds "No HIP GPUs are available" in torch afterwards (seen in
    round 6; the other order works).  No GPU: nothing to do."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:   # noqa: BLE001 -- torch is test plumbing here, never the subject
        pass
    yield


@pytest.fixture(scope="session")
def weights_bytes():
    return open(WEIGHTS, "rb").read()


@pytest.fixture(scope="session")
def golden_io():
    inp = np.fromfile(os.path.join(GOLDEN, "testing.raw"), dtype="<i2").astype(np.float32)
    ref = np.fromfile(os.path.join(GOLDEN, "reference_output.raw"), dtype="<i2")
    nfr = len(inp) // 480  # chunks_exact: the 44 trailing samples are dropped (src/lib.rs:204)
    return inp[: nfr * 480].reshape(nfr, 480), ref


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hostsim_lib():
    """The product sources under the TEST-ONLY SIMT interpreter (tests/hostsim)."""
    import build_hostsim
    from nnnoiseless_amd import _ffi
    return _ffi.Library(build_hostsim.build())


@pytest.fixture(scope="session")
def gpu_lib():
    import nnnoiseless_amd
    return nnnoiseless_amd.library()


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = float((b ** 2).sum())
    return (float(((a - b) ** 2).sum()) / den) ** 0.5 if den > 0 else float(np.abs(a - b).max())


def flip_stats(gpu_branch, gpu_out, ref, ref32):
    """The excuse for frames whose pitch-filter branch mask (`exp > g` per band, src/features.rs:227; bit 22: the silence gate) differs
    from the oracle's, on data: the three pairings of the GPU, the oracle with its FFT in f64 (the checker) and the
    oracle with its FFT in f32 (the arithmetic of the reference's own rustfft).  For each pairing the number of frames whose masks
    differ and the UNMASKED relative RMS of the audio (frame 0 dropped, as the reference's callers drop it).  The reference's own
    arithmetic against itself sets the scale: a GPU that flips about as often as f32-vs-f64 does is as close as the reference gets
    to itself.  gpu_branch / ref[...]["branch"]: [S, T] int32, outs [S, T, 480]."""
    def pair(b1, o1, b2, o2):
        return int((b1 != b2).sum()), rel_rms(o1[:, 1:], o2[:, 1:])
    fg64, rg64 = pair(gpu_branch, gpu_out, ref["branch"], ref["out"])
    fg32, rg32 = pair(gpu_branch, gpu_out, ref32["branch"], ref32["out"])
    f3264, r3264 = pair(ref32["branch"], ref32["out"], ref["branch"], ref["out"])
    return {"frames": int(gpu_branch.size), "flips_gpu_vs_f64": fg64, "flips_gpu_vs_f32": fg32, "flips_f32_vs_f64": f3264,
            "rel_rms_unmasked_gpu_vs_f64": rg64, "rel_rms_unmasked_gpu_vs_f32": rg32, "rel_rms_unmasked_f32_vs_f64": r3264}


def assert_flips_in_line(st, tag=""):
    """The GPU may flip a branch about as often as the reference's own f32 arithmetic does against the checker.  As a RATE since round 6
    (profiles/r6_parity_flip_rates.json, 3.06 M frames: GPU vs f64 14.2 flips per million on the synthetic streams [9.5, 20.3], f32 vs
    f64 11.2 [7.1, 16.9]; real audio 4.9 [1.6, 11.5] against 2.0 [0.2, 7.1]): the count must be inside the 99.9 % Poisson quantile of
    TWICE the upper end of the reference arithmetic's own interval, 34 per million -- and, as before, no more than a few times what the
    f32 oracle flipped on the same inputs when that is the larger allowance."""
    from scipy.stats import poisson
    by_rate = int(poisson.ppf(0.999, 34e-6 * st["frames"]))
    bound = max(4, by_rate, 3 * st["flips_f32_vs_f64"])
    assert st["flips_gpu_vs_f64"] <= bound, (tag, bound, st)
