#!/bin/bash
# Round-2b session A: parity suite on the fused pitch kernel, headline bench with per-kernel times, scale bench, lane variants.
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
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
timeout 300 python bench.py --no-cpu-baseline --no-also > gpurun_out/b1.json 2> gpurun_out/b1.err; echo "bench rc=$?"
show gpurun_out/b1.json 4096x48
timeout 300 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b2.json 2> gpurun_out/b2.err; echo "bench2 rc=$?"
show gpurun_out/b2.json 65536x48
for LN in 1 3; do
NNN_LANES=$LN timeout 300 python bench.py --no-cpu-baseline --no-also --no-roofline --no-tick > gpurun_out/b1_l$LN.json 2> gpurun_out/b1_l$LN.err
python -c "
import json; d=json.load(open('gpurun_out/b1_l$LN.json')); print('lanes $LN 4096x48: %.2f M' % (d['value']/1e6))"
done
