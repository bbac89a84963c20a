#!/bin/bash
# Register / LDS / spill summary of every kernel, from device-only assembly of the product sources (no GPU needed).
# usage: scripts/kernel_regs.sh [extra -D flags]      (assembly left in /tmp/nnn_regs/k.s)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/nnn_regs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -S -o /tmp/nnn_regs/k.s \
    -x hip "$R/nnnoiseless_amd/csrc/nnn_batch.hip" -I "$R/nnnoiseless_amd/csrc" '-DNNN_WEIGHTS_PATH="x"' -Wno-unused-value "$@"
python3 - <<'P'
import re, subprocess
t = open('/tmp/nnn_regs/k.s').read()
t = t[t.index('amdhsa.kernels:'):]
for blk in re.split(r'\n  - \.agpr_count', t)[1:]:
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
    name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.split('(')[0]
    print(f"{name:28s} vgpr {g('vgpr_count'):>4s}  spill {g('vgpr_spill_count'):>3s}  sgpr {g('sgpr_count'):>4s}  lds {g('group_segment_fixed_size'):>6s}  scratch {g('private_segment_fixed_size'):>4s}")
P
