#!/bin/bash
# runs tools/bench_hamming.py under rocprofv3 for every probe library given (bits) and prints the two sweep kernels' averages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for bits in "$@"; do
  if [ "$bits" = "0" ]; then unset PS_HIP_LIB; else export PS_HIP_LIB=$GRAFT_REPO_ROOT/tools/ubench/_dbg/libps_dbg$bits.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_p$bits -o p -- python tools/bench_hamming.py ${NBITS:-512} ${NQ:-10000} ${NITEMS:-59047} ${TOPK:-11} 5 > gpurun_out/probe_$bits.log 2>&1
  python - <<PY
import csv
out=[]
for r in csv.DictReader(open('gpurun_out/prof_p$bits/p_kernel_stats.csv')):
    if 'hamming_mfma_kernel' in r['Name'] or 'hamming_pipe_kernel' in r['Name']:
        out.append(('collect' if (', 1, ' in r['Name'] or ', 1>' in r['Name']) else 'bound', round(float(r['AverageNs'])/1e3,1)))
print('bits=$bits', out)
PY
done
