"""The run-time knobs the product reads from the environment (the table in include/nnn_batch.h) that no other test touches, under the
test-only SIMT interpreter: every setting gives the bits of the default.  (NNN_DEVICE needs real devices: tests/test_gpu_node.py.)"""
import numpy as np
import pytest


@pytest.mark.parametrize("env", [{"NNN_SCHED": "seq"}, {"NNN_SCHED": "lanes", "NNN_LANES": "2"}, {"NNN_SCHED": "stages"}, {"NNN_LANES": "3"}])
def test_schedule_knobs_give_the_same_bits(hostsim_lib, monkeypatch, env):
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 3, 34                                   # (calls of 32 frames or more are the ones a schedule spreads over streams)
    x = make_streams(5, S, T)
    want, want_vad = nn.BatchDenoiser(S, lib=hostsim_lib).process(x)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    bd = nn.BatchDenoiser(S, lib=hostsim_lib)      # (read at creation)
    got, vad = bd.process(x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(vad.view(np.uint32), want_vad.view(np.uint32))


def test_developer_knobs_are_not_in_the_product_build():
    """The product library reads only the documented variables: the A/B probe knobs of earlier rounds are compiled in with
    -DNNN_DEV_KNOBS alone (the interpreter build has it, the hipcc build must not)."""
    import os
    import re
    from nnnoiseless_amd.build import CSRC, LIB_PATH
    src = open(os.path.join(CSRC, "nnn_batch.hip")).read()
    product = set(re.findall(r'\bknob\("(NNN_[A-Z_0-9]+)"\)', src))
    product |= set(re.findall(r'getenv\("(NNN_[A-Z_0-9]+)"\)', open(os.path.join(CSRC, "nnn_node.cpp")).read()))
    product |= set(re.findall(r'getenv\("(NNN_[A-Z_0-9]+)"\)', open(os.path.join(CSRC, "rnnoise_capi.cpp")).read()))
    header = open(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "nnn_batch.h")).read()
    table = header[header.index(" * Environment."):header.index("Earlier rounds' A/B probe knobs")]
    assert product == set(re.findall(r"^ \*   (NNN_[A-Z_0-9]+)", table, re.M)), product     # the table IS the list
    assert len(product) <= 10
    assert not re.findall(r'(?<![_a-z])getenv\("NNN_', src)                                  # nothing reads the environment behind the table's back
    if os.path.exists(LIB_PATH):
        blob = open(LIB_PATH, "rb").read()
        for name in re.findall(r'dev_knob\("(NNN_[A-Z_0-9]+)"\)', src):
            assert name.encode() + b"\0" not in blob, name
