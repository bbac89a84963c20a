"""SURVEY.md 8(f) #3 on the CPU SIMT-interpreter build: 87-column training rows against the oracle's restatement of
src/training.rs:113-160."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "hostsim"))


@pytest.fixture(scope="module")
def hostsim_lib():
    import build_hostsim
    from nnnoiseless_amd import _ffi
    return _ffi.Library(build_hostsim.build())


def test_training_rows(hostsim_lib, oracle_mod, weights_bytes):
    from nnnoiseless_amd.training import ROW_WIDTH, TrainingFeatures
    from train_fixtures import check_rows, make_training_inputs
    sig, noise, comb, cutoff, vad = make_training_inputs(5, 70, 12)
    ref = oracle_mod.training_rows(oracle_mod.Model(weights_bytes), sig, noise, comb, cutoff, vad)
    tf = TrainingFeatures(70, lib=hostsim_lib)
    rows = np.concatenate([tf.process(sig[:, :5], noise[:, :5], comb[:, :5], cutoff[:5], vad[:5]),
                           tf.process(sig[:, 5:], noise[:, 5:], comb[:, 5:], cutoff[5:], vad[5:])])
    assert ROW_WIDTH == 87
    ref32 = oracle_mod.training_rows(oracle_mod.Model(weights_bytes, f32_fft=True), sig, noise, comb, cutoff, vad)
    check_rows(rows, ref, ref32)
    assert (ref[..., 42:64] == -1.0).any() and (ref[..., :42] == 0).all(axis=-1).any()   # both special cases occur
    tf.reset()
    again = tf.process(sig[:, :5], noise[:, :5], comb[:, :5], cutoff[:5], vad[:5])
    assert np.array_equal(again, rows[:5])


def test_training_host_call_in_chunks(hostsim_lib, monkeypatch):
    """The host entry point ships a long call in chunks of frames (uploads, kernels and downloads side by side on the GPU):
    the chunks must not show."""
    from nnnoiseless_amd.training import TrainingFeatures
    from train_fixtures import make_training_inputs
    sig, noise, comb, cutoff, vad = make_training_inputs(6, 9, 7)
    res = []
    for chunk in ("0", "3"):
        monkeypatch.setenv("NNN_HOST_CHUNK", chunk)
        res.append(TrainingFeatures(9, lib=hostsim_lib).process(sig, noise, comb, cutoff, vad))
    assert np.array_equal(res[0], res[1])
