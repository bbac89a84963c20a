#!/bin/bash
# Round-2b session C: quick parity subset, benches with per-kernel times, chained vs looped k_pitch
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -x -q -k "${K:-golden or every_stage or 1024x40 or custom or grouped or rows or nonfinite or two_frames or clone or wide or edge or repeat or patterns}" 2>&1 | tail -4
show() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d=json.load(open(sys.argv[1]))
    t = d.get('tick') or {}
    print('%s: %.2f M  tick %.2f M' % (sys.argv[2], d['value']/1e6, t.get('value', 0)/1e6), {k[2:]: (round(v['avg_us'],1), round(v['us_per_frame'],1)) for k,v in d.get('kernels', {}).items()})
except Exception as e: print('parse fail', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
for CH in 1 0; do
export NNN_PITCH_CHAIN=$CH
timeout 300 python bench.py --no-cpu-baseline --no-also > gpurun_out/b1_c$CH.json 2> gpurun_out/b1_c$CH.err
show gpurun_out/b1_c$CH.json "chain=$CH 4096x48"
timeout 300 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline --no-tick > gpurun_out/b2_c$CH.json 2> gpurun_out/b2_c$CH.err
show gpurun_out/b2_c$CH.json "chain=$CH 65536x48"
done
unset NNN_PITCH_CHAIN
for LN in 1 3; do
NNN_LANES=$LN timeout 300 python bench.py --no-cpu-baseline --no-also --no-roofline --no-tick > gpurun_out/b1_l$LN.json 2> gpurun_out/b1_l$LN.err
show gpurun_out/b1_l$LN.json "lanes=$LN 4096x48"
done
