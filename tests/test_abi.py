"""The C-ABI library: it builds for gfx950, loads, exports every symbol the headers declare, and its
GPU-free entry points (model container) behave like the reference's.  No kernel runs here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def lib():
    from nnnoiseless_amd.build import build_library
    import nnnoiseless_amd
    build_library()
    return nnnoiseless_amd.library()


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b((?:nnn|rnnoise)_[a-z0-9_]+)\s*\(", src))


def test_exports_every_declared_symbol(lib):
    from nnnoiseless_amd import _ffi
    declared = _declared("nnn_batch.h") | _declared("rnnoise.h") | _declared("nnn_train.h") | _declared("nnn_resample.h") | _declared("nnn_node.h")
    assert declared == set(_ffi.BATCH_SYMBOLS) | set(_ffi.RNNOISE_SYMBOLS) | set(_ffi.TRAIN_SYMBOLS) | set(_ffi.RESAMPLE_SYMBOLS) | set(_ffi.NODE_SYMBOLS)
    for sym in declared:
        assert hasattr(lib.L, sym), sym
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.path]).decode()
    for sym in declared:
        assert re.search(rf"\bT {sym}\b", out), sym


def test_library_holds_gfx950_code(lib):
    blob = open(lib.path, "rb").read()
    assert b"gfx950" in blob and b"k_fft_x" in blob


def test_frame_size_and_state_size(lib):
    assert lib.L.rnnoise_get_frame_size() == 480      # src/capi.rs:16-19
    assert lib.L.rnnoise_get_size() > 0


def test_model_container_rules(lib, weights_bytes):
    import nnnoiseless_amd as nn
    assert nn.RnnModel.default().shape() == [42, 24, 24, 48, 96, 22, 0, 2, 2, 2, 1, 1]
    assert nn.RnnModel.from_bytes(weights_bytes).shape()[:6] == [42, 24, 24, 48, 96, 22]
    sh = open(os.path.join(GOLDEN, "sh.rnn"), "rb").read()
    assert nn.RnnModel.from_bytes(sh).shape()[6:] == [0, 0, 2, 0, 1, 1]
    # where the reference returns None (src/rnn.rs:196-222)
    for bad in (weights_bytes[:-1], weights_bytes + b"\0", b"", bytes([41]) + weights_bytes[1:],
                weights_bytes[:2] + b"\x03" + weights_bytes[3:], bytes([0x80]) + weights_bytes[1:]):
        assert nn.RnnModel.from_bytes(bad) is None


def test_model_clone_is_an_independent_copy(lib, weights_bytes):
    """`RnnModel: Clone` (src/rnn.rs:54) through nnn_model_clone: same parameters, its own storage -- the copy outlives the
    original; a NULL model gives NULL and an error text, not a crash."""
    import copy
    import nnnoiseless_amd as nn
    sh = nn.RnnModel.from_bytes(open(os.path.join(GOLDEN, "sh.rnn"), "rb").read())
    c = sh.clone()
    assert c._h != sh._h and c.shape() == sh.shape()
    want = sh.shape()
    del sh
    assert c.shape() == want and copy.copy(c).shape() == want
    assert not lib.L.nnn_model_clone(None)
    assert b"null model" in lib.L.nnn_last_error()


def test_rust_facade_keeps_clone_and_sync():
    """The Rust façade (never compiled here: no toolchain) must keep what the reference promises: `RnnModel: Clone`
    (src/rnn.rs:54) and `DenoiseState: Clone + Send + Sync` (src/denoise.rs:36,125).  A text check is all this image allows."""
    src = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    for needle in ("impl Clone for RnnModel", "fn nnn_model_clone(", "unsafe impl Sync for BatchDenoiser", "unsafe impl Send for BatchDenoiser",
                   # the lifetime and the three constructors' exact signatures (src/denoise.rs:37,44-74)
                   "#[derive(Clone)]\npub struct DenoiseState<'model> {", "_model: std::marker::PhantomData<&'model RnnModel>",
                   "impl DenoiseState<'static> {", "pub fn new() -> Box<DenoiseState<'static>>",
                   "pub fn from_model(model: RnnModel) -> Box<DenoiseState<'static>>", "impl<'model> DenoiseState<'model> {",
                   "pub fn with_model(model: &'model RnnModel) -> Box<DenoiseState<'model>>",
                   "pub fn process_frame(&mut self, output: &mut [f32], input: &[f32]) -> f32",
                   # DenoiseSignal (src/signal.rs:29-138) behind the dasp feature, as in the reference
                   "pub struct DenoiseSignal<'model, S: Signal> {", "pub fn new(input: S) -> DenoiseSignal<'static, S>",
                   "pub fn with_model(input: S, model: &'model RnnModel) -> DenoiseSignal<'model, S>",
                   "pub fn from_model(input: S, model: RnnModel) -> DenoiseSignal<'static, S>",
                   "impl<'model, S: Signal> Signal for DenoiseSignal<'model, S> {"):
        assert needle in src, needle


def test_model_from_file_closes_and_parses(lib, tmp_path, weights_bytes):
    """rnnoise_model_from_file takes ownership of the FILE* (src/capi.rs:93-94)."""
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    lib.L.rnnoise_model_from_file.restype = C.c_void_p
    lib.L.rnnoise_model_from_file.argtypes = [C.c_void_p]
    good = tmp_path / "m.rnn"
    good.write_bytes(weights_bytes)
    m = lib.L.rnnoise_model_from_file(libc.fopen(str(good).encode(), b"rb"))
    assert m
    lib.L.rnnoise_model_free(m)
    bad = tmp_path / "bad.rnn"
    bad.write_bytes(weights_bytes[:1000])
    assert not lib.L.rnnoise_model_from_file(libc.fopen(str(bad).encode(), b"rb"))


def test_no_cpu_fallback_without_gpu(lib):
    """Without a usable HIP device the product fails loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import nnnoiseless_amd as nn
    with pytest.raises(RuntimeError):
        nn.BatchDenoiser(4)
    lib.L.rnnoise_create.restype = C.c_void_p
    assert not lib.L.rnnoise_create(None)


def test_product_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nnnoiseless_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() and "hostsim" not in text.lower(), os.path.join(dirpath, f)


def test_reference_demo_compiles_against_our_header(tmp_path):
    """The reference's C demo (test_data/rnnoise_demo.c) builds unmodified against include/rnnoise.h.
    Only where the reference tree is mounted (build container)."""
    demo = "/root/reference/test_data/rnnoise_demo.c"
    if not os.path.exists(demo):
        pytest.skip("reference tree not present")
    subprocess.check_call(["gcc", "-c", "-I", os.path.join(ROOT, "include"), demo, "-o", str(tmp_path / "demo.o")])


def test_cpp_mirror_header_compiles(tmp_path):
    """include/nnnoiseless.hpp (header-only C++ mirror of RnnModel / DenoiseState / BatchDenoiser / TrainingFeatures)."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "nnnoiseless.hpp"\nint main() { return 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"), str(src)])


def test_bench_knows_every_kernel(lib):
    """bench.py's roofline table must have an entry for every kernel the library times (a missing key would only show
    up on the GPU box)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lib.L.nnn_batch_kernel_name.restype = C.c_char_p
    names = {lib.L.nnn_batch_kernel_name(k).decode() for k in range(lib.L.nnn_batch_num_kernels())}
    assert names and names <= set(bench.KERNEL_BYTES), names - set(bench.KERNEL_BYTES)
