#!/bin/bash
# A/B of library variants (nnnoiseless_amd/lib/variants/<name>.so through NNN_LIBRARY, built by scripts/build_variant.sh):
# headline + config 2, per-kernel times.   VARIANTS="default fft5 ..." bash scripts/gpu_ab.sh
set -u
mkdir -p gpurun_out
for V in ${VARIANTS:-default}; do
  if [ "$V" = default ]; then unset NNN_LIBRARY; else export NNN_LIBRARY=$PWD/nnnoiseless_amd/lib/variants/$V.so; fi
  for C in ${CONFIGS:-1 2}; do
    timeout 300 python bench.py --config $C --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-tick > gpurun_out/ab.json 2>gpurun_out/ab.err
    python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab.json'))
    print('$V config $C: %.2f M' % (d['value']/1e6), {k[2:]: round(v['us_per_frame'],1) for k,v in d['kernels'].items()})
except Exception as e: print('parse fail', e); print(open('gpurun_out/ab.err').read()[-800:])
PY
  done
done
