"""model tooling -- RNNoise text -> .rnn conversion (train/convert_rnnoise.py) and several models
resident in one batch.  CPU: host logic + the SIMT-interpreter build of the kernels against the oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, rel_rms

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "hostsim"))


@pytest.fixture(scope="module")
def hostsim_lib():
    import build_hostsim
    from nnnoiseless_amd import _ffi
    return _ffi.Library(build_hostsim.build())


def _as_text(blob, sep=" "):
    """The text form a .rnn file came from: signed integers after the rnnoise-nu header line."""
    vals = np.frombuffer(blob, dtype=np.int8)
    return "rnnoise-nu model file version 1\n" + sep.join(str(int(v)) for v in vals) + "\n"


def test_convert_text_python_and_c_agree(hostsim_lib):
    from nnnoiseless_amd.convert import convert_rnnoise_text
    sh = open(os.path.join(GOLDEN, "sh.rnn"), "rb").read()
    text = _as_text(sh, sep="\n ")
    assert convert_rnnoise_text(text) == sh
    # integers outside the i8 range wrap modulo 256 (python's non-negative modulo), '+' signs and CRLF are fine
    odd = "  rnnoise-nu model file version 1 \r\n 300 -1\t-129 +5 -256\r\n0"
    assert convert_rnnoise_text(odd) == bytes([44, 255, 127, 5, 0, 0])
    L = hostsim_lib.L
    for t, want in ((text, sh), (odd, bytes([44, 255, 127, 5, 0, 0]))):
        raw = t.encode()
        n = L.nnn_convert_rnnoise_text(raw, len(raw), None, 0)
        assert n == len(want)
        out = (C.c_uint8 * n)()
        assert L.nnn_convert_rnnoise_text(raw, len(raw), out, n) == n
        assert bytes(out) == want
        assert L.nnn_convert_rnnoise_text(raw, len(raw), out, n - 1) == -1          # too little room
    for bad in ("rnnoise model file version 1\n1 2 3", "rnnoise-nu model file version 1\n1 x 3", "rnnoise-nu model file version 1\n1 2.5"):
        with pytest.raises(ValueError):
            convert_rnnoise_text(bad)
        assert L.nnn_convert_rnnoise_text(bad.encode(), len(bad), None, 0) == -1
    assert not L.nnn_model_from_rnnoise_text(b"rnnoise-nu model file version 1\n1 2 3", 38)   # converts, wrong shape


def test_convert_cli(tmp_path):
    sh = open(os.path.join(GOLDEN, "sh.rnn"), "rb").read()
    src, dst = tmp_path / "m.txt", tmp_path / "m.rnn"
    src.write_text(_as_text(sh))
    r = subprocess.run([sys.executable, "-m", "nnnoiseless_amd.convert", str(src), str(dst)], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "Converted" in r.stdout and dst.read_bytes() == sh
    src.write_text("something else\n1 2 3")
    r = subprocess.run([sys.executable, "-m", "nnnoiseless_amd.convert", str(src), str(dst)], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 1 and "Unexpected input file format" in r.stdout
    r = subprocess.run([sys.executable, "-m", "nnnoiseless_amd.convert", str(src)], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 1 and "USAGE" in r.stdout


def test_text_model_and_grouped_batch(hostsim_lib, oracle_mod, weights_bytes):
    """Two models resident at once: streams 0..63 on the built-in model, 64..69 on the converted rnnoise-nu model."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    sh = open(os.path.join(GOLDEN, "sh.rnn"), "rb").read()
    m_sh = nn.RnnModel.from_rnnoise_text(_as_text(sh), lib=hostsim_lib)
    assert m_sh is not None and m_sh.shape() == nn.RnnModel.from_bytes(sh, lib=hostsim_lib).shape()
    x = make_streams(21, 70, 3)
    bd = nn.BatchDenoiser(70, lib=hostsim_lib, groups=[(None, 64), (m_sh, 6)])
    out, vad = bd.process(x)
    ref0 = oracle_mod.run_streams(oracle_mod.Model(weights_bytes), x[:64])
    ref1 = oracle_mod.run_streams(oracle_mod.Model(sh), x[64:])
    assert rel_rms(out[:64, 1:], ref0["out"][:, 1:]) < 1e-5 and rel_rms(out[64:, 1:], ref1["out"][:, 1:]) < 1e-5
    assert np.abs(vad.T[:64] - ref0["vad"]).max() < 1e-4 and np.abs(vad.T[64:] - ref1["vad"]).max() < 1e-4
    assert rel_rms(out[64:, 1:], ref0["out"][:6, 1:]) > 1e-3          # the models really differ
    with pytest.raises(RuntimeError):
        nn.BatchDenoiser(70, lib=hostsim_lib, groups=[(None, 6), (m_sh, 64)])   # a non-final group must fill whole tiles


def test_grouped_models_of_different_widths(hostsim_lib, oracle_mod, weights_bytes):
    """A narrower and a wider synthetic model next to the built-in one: per-model plans, shared state arrays."""
    import nnnoiseless_amd as nn
    from model_fixtures import make_model
    from nnnoiseless_amd.synthetic import make_streams
    small, wide = make_model(16, 20, 40, 72, seed=1), make_model(32, 28, 56, 120, acts=(0, 0, 2, 0, 1, 1), seed=2)
    ms, mw = nn.RnnModel.from_bytes(small, lib=hostsim_lib), nn.RnnModel.from_bytes(wide, lib=hostsim_lib)
    assert ms is not None and mw is not None
    biggest = make_model(42, 43, 42, 127, seed=3)      # the widest the i8 headers allow (inputs <= 127 everywhere)
    mb = nn.RnnModel.from_bytes(biggest, lib=hostsim_lib)
    assert mb is not None
    x = make_streams(31, 64 + 64 + 64 + 5, 3)
    bd = nn.BatchDenoiser(197, lib=hostsim_lib, groups=[(ms, 64), (None, 64), (mb, 64), (mw, 5)])
    out, vad = bd.process(x)
    for sl, blob in ((slice(0, 64), small), (slice(64, 128), weights_bytes), (slice(128, 192), biggest), (slice(192, 197), wide)):
        ref = oracle_mod.run_streams(oracle_mod.Model(blob), x[sl])
        assert rel_rms(out[sl, 1:], ref["out"][:, 1:]) < 1e-5
        assert np.abs(vad.T[sl] - ref["vad"]).max() < 1e-4
