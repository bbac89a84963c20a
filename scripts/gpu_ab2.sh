#!/bin/bash
# A/B of library variants at the headline batch, 48 frames per call and 384, plus config 2
set -u
mkdir -p gpurun_out
for V in ${VARIANTS:-default}; do
  if [ "$V" = default ]; then unset NNN_LIBRARY; else export NNN_LIBRARY=$PWD/nnnoiseless_amd/lib/variants/$V.so; fi
  echo "== $V"
  FPSS="48 384" bash scripts/gpu_fps.sh
  timeout 300 python bench.py --config 2 --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-tick --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config 2: %.2f M' % (d['value']/1e6))"
done
