#!/bin/bash
# PMC passes over the Hamming scan alone (tools/bench_hamming.py): usage tools/hm_pmc.sh TAG [NBITS]   (through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-x}; NB=${2:-256}
OUT=gpurun_out/hm_pmc_$TAG
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_IFETCH" \
           "SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "hamming_pipe|hamming_mfma" --output-format csv -d $OUT/p$i -o p -- python tools/bench_hamming.py $NB 10000 59047 11 3 > $OUT/p$i.log 2>&1
done
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
cat $OUT/summary.txt
