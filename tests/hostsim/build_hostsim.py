"""Builds the TEST-ONLY SIMT-interpreter build of the product sources (g++, no GPU, no hipcc).

The same nnnoiseless_amd/csrc/*.hip files are compiled against tests/hostsim/hip/hip_runtime.h, a
stand-in runtime that runs each workgroup's threads as fibers on the CPU.  Used by the `not gpu`
tests to check kernel logic against the oracle where no MI355X is present.  Never used by the package.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "nnnoiseless_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libnnn_hostsim.so")


def build(force=False):
    srcs = [os.path.join(CSRC, s) for s in ("nnn_batch.hip", "nnn_resample.hip", "nnn_model.cpp", "rnnoise_capi.cpp", "nnn_node.cpp")]
    srcs.append(os.path.join(HERE, "hostsim.cpp"))
    deps = srcs + [os.path.join(CSRC, d) for d in ("nnn_kernels.hip", "nnn_back.hip", "nnn_layout.h", "nnn_model.h")] + [os.path.abspath(__file__)]
    deps += [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "nnn_mfma.h"), os.path.join(HERE, "hostsim.cpp")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    weights = os.path.join(ROOT, "nnnoiseless_amd", "data", "weights.rnn")
    cmd = ["g++", "-O2", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-DNNN_DEV_KNOBS",
           "-I", HERE, "-I", CSRC, f'-DNNN_WEIGHTS_PATH="{weights}"'] + os.environ.get("NNN_HOSTSIM_DEFINES", "").split() + ["-x", "c++"] + srcs + ["-o", OUT]
    # (NNN_HOSTSIM_DEFINES: build knobs of the product sources for a one-off check under the interpreter, e.g. -DNNN_FFT_LANE_TW=1)
    subprocess.check_call(cmd)
    return OUT
