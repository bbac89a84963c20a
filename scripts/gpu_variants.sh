#!/bin/bash
# GROUP x LANES experiment: experimental builds under nnnoiseless_amd/lib/variants (made on the dev box with
# -DNNN_GROUP / -DNNN_LANES), selected through NNN_LIBRARY.
set -u
mkdir -p gpurun_out
for L in nnnoiseless_amd/lib/libnnnoiseless_mi355x.so nnnoiseless_amd/lib/variants/*.so; do
  for F in 48 96; do
    NNN_LIBRARY=$PWD/$L timeout 300 python bench.py --frames-per-step $F --steps $((1440 / F)) --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $L) frames/step=$F: %.2f M' % (d['value']/1e6))"
  done
done
