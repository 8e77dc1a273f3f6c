#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = "base" ]; then unset PS_HIP_LIB; else export PS_HIP_LIB=$GRAFT_REPO_ROOT/tools/ubench/_dbg/libps_ws$v.so; fi
  python tools/ws_knockout_run.py 2>&1 | grep -v "Warning\|amdgpu" | tail -1
done
