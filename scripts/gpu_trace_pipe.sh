#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/trace_pipe
timeout 600 rocprofv3 --kernel-trace -d $O/trace_pipe -o tr -- python $R/bench.py --streams 4096 --frames-per-step 16 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/trace_pipe.json 2> $O/trace_pipe.err
tail -1 $O/trace_pipe.err | cut -c1-100; cat $O/trace_pipe.json | cut -c1-200
