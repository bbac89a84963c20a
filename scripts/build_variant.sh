#!/bin/bash
# Developer tool: an experimental build of the library with extra -D flags, under nnnoiseless_amd/lib/variants/<name>.so
# (selected at run time through NNN_LIBRARY; see scripts/gpu_ab.sh).   usage: build_variant.sh name -DNNN_X=1 ...
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/nnnoiseless_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-unused-value -DNNN_DEV_KNOBS "$@" -I $R/nnnoiseless_amd/csrc \
  -DNNN_WEIGHTS_PATH="\"$R/nnnoiseless_amd/data/weights.rnn\"" -x hip $R/nnnoiseless_amd/csrc/nnn_batch.hip $R/nnnoiseless_amd/csrc/nnn_resample.hip \
  $R/nnnoiseless_amd/csrc/nnn_model.cpp $R/nnnoiseless_amd/csrc/rnnoise_capi.cpp $R/nnnoiseless_amd/csrc/nnn_node.cpp -o $R/nnnoiseless_amd/lib/variants/$name.so 2>/dev/null
echo built $name
