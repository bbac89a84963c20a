#!/bin/bash
# Developer tool: the library as of a commit, under nnnoiseless_amd/lib/variants/<name>.so (same-box A/B against the working tree).
# usage: build_variant_from_commit.sh name commit|WORK [-D...]      (WORK = the working tree as it is)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; commit=$2; shift 2
T=$(mktemp -d)
mkdir -p $T/nnnoiseless_amd/csrc $T/include $R/nnnoiseless_amd/lib/variants
if [ "$commit" = WORK ]; then cp $R/nnnoiseless_amd/csrc/* $T/nnnoiseless_amd/csrc/; cp $R/include/* $T/include/
else for f in $(git -C $R ls-tree --name-only $commit nnnoiseless_amd/csrc/ include/); do git -C $R show $commit:$f > $T/$f; done; fi
cd $T/nnnoiseless_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-unused-value "$@" -I . \
  -DNNN_WEIGHTS_PATH="\"$R/nnnoiseless_amd/data/weights.rnn\"" -x hip nnn_batch.hip nnn_resample.hip nnn_model.cpp rnnoise_capi.cpp $( [ -f nnn_node.cpp ] && echo nnn_node.cpp ) -o $R/nnnoiseless_amd/lib/variants/$name.so 2>/dev/null
rm -rf $T
echo built $name from $commit
