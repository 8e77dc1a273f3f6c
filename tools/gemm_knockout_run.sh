#!/bin/bash
# runs tools/gemm_probe.py once per probe library (suffixes of tools/ubench/_dbg/libps_gm*.so; "base" = the shipped library)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = "base" ]; then unset PS_HIP_LIB; else export PS_HIP_LIB=$GRAFT_REPO_ROOT/tools/ubench/_dbg/libps_gm$v.so; fi
  echo "== variant $v"
  python tools/gemm_probe.py --rounds 3 2>&1 | grep -v Warning
done
