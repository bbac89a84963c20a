#!/bin/bash
# The evidence session of a round: full GPU parity suite, then scripts/gpu_final_profile.sh (driver-style default bench, rocprofv3 kernel
# stats, PMC traffic at 4096 and 65536 streams, SQ counters at 65536, overlap trace, single-stream latency), k_pitch phase stamps.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
bash scripts/gpu_final_profile.sh
STREAMS=65536 bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -8
bash scripts/gpu_stamps_pitch.sh 2>&1 | tail -6 | tee gpurun_out/pitch_stamps.txt
