#!/bin/bash
# Round-2 first session: parity suite on the new 7-stage pipeline, headline bench with per-kernel times, scale bench.
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-also > gpurun_out/b1.json 2> gpurun_out/b1.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/b1.json'))
    print('4096x48: %.2f M  tick %.2f M' % (d['value']/1e6, d['tick']['value']/1e6), {k[2:]: (round(v['avg_us'],1), round(v['us_per_frame'],1)) for k,v in d['kernels'].items()})
except Exception as e: print('parse fail', e); print(open('gpurun_out/b1.err').read()[-1500:])
PY
for SCH in seq stages; do
NNN_SCHED=$SCH timeout 300 python bench.py --no-cpu-baseline --no-also --no-roofline > gpurun_out/b1_$SCH.json 2> gpurun_out/b1_$SCH.err
python -c "
import json; d=json.load(open('gpurun_out/b1_$SCH.json')); print('$SCH 4096x48: %.2f M' % (d['value']/1e6))"
done
timeout 300 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b2.json 2> gpurun_out/b2.err; echo "bench2 rc=$?"
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/b2.json'))
    print('65536x48: %.2f M  tick %.2f M' % (d['value']/1e6, d['tick']['value']/1e6), {k[2:]: (round(v['avg_us'],1), round(v['us_per_frame'],1)) for k,v in d['kernels'].items()})
except Exception as e: print('parse fail', e); print(open('gpurun_out/b2.err').read()[-1500:])
PY
