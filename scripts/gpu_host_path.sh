#!/bin/bash
# PCIe-inclusive rates of the host-buffer entry points (pageable / page-locked buffers, f32 / packed int16).
# NNN_HOST_CHUNK=0 is the call in one piece (round 2a's behaviour), unset = the library's choice of chunk length.
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -x -q -k "chunks_and_pinned or interleaved or packed_i16" 2>&1 | tail -3
{
echo "== one piece (NNN_HOST_CHUNK=0), 4096 streams x 48 frames"; NNN_HOST_CHUNK=0 PYTHONPATH=. timeout 300 python scripts/host_path_rate.py 4096 48 6 2>&1 | grep frames/s
for cfg in "4096 48 6" "4096 192 3" "16384 48 4"; do
  set -- $cfg
  echo "== chunked, $1 streams x $2 frames"; PYTHONPATH=. timeout 300 python scripts/host_path_rate.py $1 $2 $3 2>&1 | grep frames/s
done
} | tee gpurun_out/host_path.txt
{
echo "== training rows, one piece (NNN_HOST_CHUNK=0), 4096 triples x 48 frames"; NNN_HOST_CHUNK=0 PYTHONPATH=. timeout 300 python scripts/train_host_rate.py 4096 48 4 2>&1 | grep rows/s
echo "== training rows, chunked, 4096 triples x 48 frames"; PYTHONPATH=. timeout 300 python scripts/train_host_rate.py 4096 48 4 2>&1 | grep rows/s
} | tee -a gpurun_out/host_path.txt
