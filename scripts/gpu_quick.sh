#!/bin/bash
# Quick check: a subset of the parity suite, headline and config-2 benches with per-kernel times, k_pitch phase stamps
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -x -q -k "${K:-golden or every_stage or 1024x40 or custom or grouped or rows or nonfinite or two_frames or clone or wide or edge}" 2>&1 | tail -4
show() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d=json.load(open(sys.argv[1]))
    t = d.get('tick') or {}
    print('%s: %.2f M  tick %.2f M' % (sys.argv[2], d['value']/1e6, t.get('value', 0)/1e6), {k[2:]: (round(v['avg_us'],1), round(v['us_per_frame'],1)) for k,v in d['kernels'].items()})
except Exception as e: print('parse fail', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
timeout 300 python bench.py --no-cpu-baseline --no-also > gpurun_out/b1.json 2> gpurun_out/b1.err
show gpurun_out/b1.json 4096x48
timeout 300 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b2.json 2> gpurun_out/b2.err
show gpurun_out/b2.json 65536x48
bash scripts/gpu_stamps_pitch.sh 2>&1 | tail -6
