set -u
mkdir -p gpurun_out; rm -rf gpurun_out/prof
timeout 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof.log" 2>&1; echo "rocprof rc=$?"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof -name '*_results.db' | head -1)
python scripts/rocpd_kernel_stats.py "$DB" > gpurun_out/kernel_stats.md; head -14 gpurun_out/kernel_stats.md
find gpurun_out/prof -name '*.db' -delete
