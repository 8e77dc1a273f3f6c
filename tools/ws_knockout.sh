#!/bin/bash
# experiment variants of the walk sampler (csrc/walk_sample.hip, PS_WS_DEBUG bits) into tools/ubench/_dbg/ (results WRONG by design)
set -e
cd "$(dirname "$0")/../movie-recommendation-engine_amd/csrc"
mkdir -p ../../tools/ubench/_dbg
for bits in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPS_WS_DEBUG=$bits -c walk_sample.hip -o ../../tools/ubench/_dbg/ws_$bits.o 2>/dev/null
  objs=$(ls _obj/*.o | grep -v walk_sample)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/_dbg/libps_ws$bits.so $objs ../../tools/ubench/_dbg/ws_$bits.o
done
