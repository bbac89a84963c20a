#!/bin/bash
# Builds scripts/ubench/dft_mfma_probe.hip for gfx950 (-> scripts/ubench/_build/dft_probe, git-ignored, travels with gpurun) and keeps the
# device assembly in /tmp/nnn_probe/probe.s.  HOSTSIM=1: the interpreter build instead (-> /tmp/nnn_probe/probe_hs), run with no GPU.
set -e
cd "$(dirname "$0")/../.."
mkdir -p scripts/ubench/_build /tmp/nnn_probe
if [ -n "$HOSTSIM" ]; then
    g++ -O2 -g -std=c++17 -ffp-contract=off -Wno-unknown-pragmas -DNNN_PROBE_HOSTSIM $EXTRA -I tests/hostsim -I nnnoiseless_amd/csrc -I scripts/ubench \
        -x c++ scripts/ubench/dft_mfma_probe.hip tests/hostsim/hostsim.cpp -o /tmp/nnn_probe/probe_hs
    exec /tmp/nnn_probe/probe_hs
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value $EXTRA -I nnnoiseless_amd/csrc -I scripts/ubench \
    scripts/ubench/dft_mfma_probe.hip -o scripts/ubench/_build/dft_probe
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value $EXTRA -I nnnoiseless_amd/csrc -I scripts/ubench \
    --cuda-device-only -S scripts/ubench/dft_mfma_probe.hip -o /tmp/nnn_probe/probe.s
grep -E "^\s+\.amdhsa_kernel _Z7k_probe|amdhsa_next_free_vgpr|amdhsa_accum_offset|amdhsa_group_segment_fixed_size|amdhsa_private_segment_fixed_size" /tmp/nnn_probe/probe.s | \
    awk '/amdhsa_kernel/ {k = ($2 ~ /k_probe/)} k {print}'
