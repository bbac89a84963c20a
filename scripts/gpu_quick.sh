#!/bin/bash
# Quick check: a subset of the parity suite, headline bench with per-kernel times, config 2.
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "${K:-golden or every_stage or 1024x40 or custom or grouped or rows or nonfinite or two_frames or clone or wide}" 2>&1 | tail -${TAIL:-6}
timeout 300 python bench.py --no-cpu-baseline --no-also > gpurun_out/q1.json 2>gpurun_out/q1.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/q1.json'))
    print('4096x48: %.2f M  tick %.2f M' % (d['value']/1e6, d['tick']['value']/1e6), {k[2:]: (round(v['avg_us'],1), round(v['us_per_frame'],1)) for k,v in d['kernels'].items()})
except Exception as e: print('parse fail', e); print(open('gpurun_out/q1.err').read()[-1500:])
PY
timeout 300 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/q2.json 2>gpurun_out/q2.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/q2.json'))
    print('65536x48: %.2f M  tick %.2f M' % (d['value']/1e6, d['tick']['value']/1e6), {k[2:]: (round(v['avg_us'],1), round(v['us_per_frame'],1)) for k,v in d['kernels'].items()})
except Exception as e: print('parse fail', e); print(open('gpurun_out/q2.err').read()[-1500:])
PY
