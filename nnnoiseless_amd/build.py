"""Build the gfx950 library in-tree with hipcc (no cmake, no JIT cache: the .so must travel)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libnnnoiseless_mi355x.so")
WEIGHTS = os.path.join(HERE, "data", "weights.rnn")
SOURCES = ["nnn_batch.hip", "nnn_resample.hip", "nnn_model.cpp", "rnnoise_capi.cpp", "nnn_node.cpp"]
HP_SOURCE = "nnn_hp.hip"
DEPS = SOURCES + [HP_SOURCE, "nnn_kernels.hip", "nnn_back.hip", "nnn_layout.h", "nnn_model.h", "nnn_mfma.h"]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, d) for d in DEPS] + [WEIGHTS, os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> nnnoiseless_amd/lib/libnnnoiseless_mi355x.so"""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value", "-I", CSRC,
              f'-DNNN_WEIGHTS_PATH="{WEIGHTS}"']
    # the two high-pass kernels in a unit of their own, with the SLP pairing the rest of the library is built without (nnn_kernels.hip, at k_hp2)
    # (per-process temporaries, the library moved into place at the end: two builds at once -- a test run beside a manual build -- do not
    # see each other's half-written files)
    hp_obj = os.path.join(LIB_DIR, f"nnn_hp.{os.getpid()}.o")
    tmp_lib = os.path.join(LIB_DIR, f"libnnnoiseless_mi355x.{os.getpid()}.so")
    cmds = [common + ["-c", "-x", "hip", os.path.join(CSRC, HP_SOURCE), "-o", hp_obj],
            common + ["-fno-slp-vectorize", "-DNNN_HP_EXTERN", "-shared", "-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES]
            + ["-x", "none", hp_obj, "-o", tmp_lib]]
    try:
        for cmd in cmds:
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        os.replace(tmp_lib, LIB_PATH)
    finally:
        for f in (hp_obj, tmp_lib):
            if os.path.exists(f):
                os.remove(f)
    return LIB_PATH
