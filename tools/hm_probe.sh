#!/bin/bash
# Builds experiment variants of libpinsage_hip.so with parts of the MFMA Hamming sweep switched off
# (csrc/hamming_mfma.hip, PS_HM_DEBUG bits) into tools/ubench/_dbg/ (git-ignored, travels with gpurun).
set -e
cd "$(dirname "$0")/../movie-recommendation-engine_amd/csrc"
mkdir -p ../../tools/ubench/_dbg
for bits in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-honor-nans -DPS_HM_DEBUG=$bits -c hamming_mfma.hip -o ../../tools/ubench/_dbg/hm_$bits.o 2>/dev/null
  objs=$(ls _obj/*.o | grep -v hamming_mfma | grep -v amdgcn)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/_dbg/libps_dbg$bits.so $objs ../../tools/ubench/_dbg/hm_$bits.o
done
