"""The matrix-core DFT of round 5's probe (scripts/ubench/nnn_dft_mfma.h; measured and NOT adopted, see
profiles/r5_dft_mfma_probe.txt) under the test-only SIMT interpreter: fragment layouts, index maps, plane splits and the scaling
give a 480-point DFT that agrees with a double-precision one, for both operand schemes, next to today's fft480_regs."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_matrix_dft_against_double_precision(tmp_path):
    env = dict(os.environ, HOSTSIM="1")
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "ubench", "build_dft_probe.sh")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    blocks = re.split(r"\n(?=\S)", out.stdout.strip())
    errs = {}
    for blk in blocks:
        name = blk.split("\n")[0].split()[0]
        errs[name] = [float(x) for x in re.findall(r"rel-rms error (\S+)", blk)]
    assert set(errs) == {"fft480_regs", "dft480_mfma<DftF16>", "dft480_mfma<DftBf16>"}, out.stdout
    # (the interpreter's MFMA sums its 32 products one fused multiply-add at a time: a pessimistic stand-in for the hardware's accumulation)
    assert max(errs["fft480_regs"]) < 2e-7
    assert max(errs["dft480_mfma<DftBf16>"]) < 3e-7
    assert max(errs["dft480_mfma<DftF16>"][:2]) < 3e-7 and errs["dft480_mfma<DftF16>"][2] < 1e-6   # (third set: 100 dB of dynamic range, the scheme's floor)
