#!/bin/bash
# builds and runs scripts/repro_graph_replay.cpp under a timeout (a hang or crash of the runtime must not take the box with it)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/repro_graph_replay.cpp -o /tmp/repro_graph || exit 1
for args in "200 4" "2000 8" "5000 16"; do
  timeout 60 /tmp/repro_graph $args; echo "rc=$? ($args)"
done
