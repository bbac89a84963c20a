#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (headline + packed int16 boundary), rocprofv3 kernel stats.
# Usage on the dev box:  gpurun --timeout 1500 -- 'bash scripts/gpu_session.sh'
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/bench.json
timeout 300 python bench.py --pcm i16 --no-cpu-baseline > gpurun_out/bench_i16.json 2> gpurun_out/bench_i16.err; echo "bench i16 rc=$?"; cut -c1-400 gpurun_out/bench_i16.json
timeout 300 python bench.py --pcm i16 --channels 2 --no-cpu-baseline --no-roofline > gpurun_out/bench_i16_stereo.json 2> gpurun_out/bench_i16_stereo.err; echo "bench i16x2 rc=$?"; cut -c1-300 gpurun_out/bench_i16_stereo.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof.log" 2>&1; echo "rocprof rc=$?"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof -name '*_results.db' | head -1)
[ -n "$DB" ] && python scripts/rocpd_kernel_stats.py "$DB" > gpurun_out/kernel_stats.md 2>&1 && head -20 gpurun_out/kernel_stats.md
find gpurun_out/prof -name '*.db' -size +20M -delete
for S in 4096 16384; do timeout 300 python bench.py --workload train --streams $S --steps 20 --warmup 3 > gpurun_out/bench_train_$S.json 2>gpurun_out/bench_train_$S.err; cut -c1-200 gpurun_out/bench_train_$S.json; done
for S in 16384 65536; do
  timeout 300 python bench.py --streams $S --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$S.json 2>gpurun_out/bench_$S.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$S.json'))
print('S=$S: %.2f M, tick %.2f M' % (d['value']/1e6, d['tick']['value']/1e6), {k[2:]: round(v['avg_us']) for k,v in d['kernels'].items()})"
done
bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -3
bash scripts/gpu_trace.sh 2>&1 | tail -8
