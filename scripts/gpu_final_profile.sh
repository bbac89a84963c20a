#!/bin/bash
# The round's evidence in one session: driver-style default bench (JSON line with roofline, cpu_baseline, configs[2] and [4]),
# rocprofv3 --kernel-trace --stats summary of the same command, PMC traffic at 4096 streams, SQ counters at 65536 streams,
# overlap trace, single-stream latency.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out; rm -rf gpurun_out/prof
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-260 gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-also > "$R/gpurun_out/prof.log" 2>&1; echo "rocprof rc=$?"
cd "$R"
DB=$(find gpurun_out/prof -name '*_results.db' | head -1)
python scripts/rocpd_kernel_stats.py "$DB" > gpurun_out/kernel_stats.md; head -10 gpurun_out/kernel_stats.md
find gpurun_out/prof -name '*.db' -delete
bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -10
bash scripts/gpu_trace.sh 2>&1 | head -8
bash scripts/gpu_scale_profile.sh 2>&1 | tail -11
timeout 300 python scripts/gpu_single_stream.py 2>&1 | tail -4
