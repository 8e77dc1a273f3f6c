#!/bin/bash
# Builds experiment variants of libpinsage_hip.so with parts of gemm_f32_kernel switched off (csrc/dense_mfma.hip,
# PS_GEMM_DEBUG bits) into tools/ubench/_dbg/ (git-ignored, travels with gpurun).  Results are WRONG by design: rates only.
set -e
cd "$(dirname "$0")/../movie-recommendation-engine_amd/csrc"
mkdir -p ../../tools/ubench/_dbg
for bits in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPS_GEMM_DEBUG=$bits -c dense_mfma.hip -o ../../tools/ubench/_dbg/gm_$bits.o 2>/dev/null
  objs=$(ls _obj/*.o | grep -v dense_mfma)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/_dbg/libps_gm$bits.so $objs ../../tools/ubench/_dbg/gm_$bits.o
done
