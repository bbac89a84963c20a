#!/bin/bash
# GPU parity suite only
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -${TAIL:-40} gpurun_out/pytest_gpu.log
