#!/bin/bash
# Collects the round's profile evidence on the GPU box into gpurun_out/profiles_$1/ (copy what is judged into profiles/):
#   <tag>/kernel_stats.csv + bench.json : rocprofv3 --kernel-trace --stats over bench.py for the default config (both RNG
#                                          modes) and BASELINE configs 2 / 3
#   default/pmc_traffic.txt             : FETCH_SIZE / WRITE_SIZE, separate --pmc passes (kernel-trace only)
#   default/pmc_mfma.txt                : MFMA / VALU busy counters
# usage: tools/profile_round.sh r02   (run through gpurun; rocprofv3 gets the program itself after --)
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
run_trace() {   # tag, bench args...
  tag=$1; shift
  mkdir -p $OUT/$tag
  # the plain run carries cpu_baseline + parity_check (the whole catalogue through the C oracle, ~3 s); the profiled run does not
  python bench.py --steps 20 --warmup 3 "$@" > $OUT/$tag/bench.json 2> $OUT/$tag/bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag/raw -o t -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/$tag/bench_under_rocprof.json 2> $OUT/$tag/rocprof.err
  cp $OUT/$tag/raw/t_kernel_stats.csv $OUT/$tag/kernel_stats.csv
  rm -rf $OUT/$tag/raw
  echo "== $tag"; python tools/show_bench.py $OUT/$tag/bench.json | head -12
}
run_trace default
run_trace default_philox --rng philox
run_trace config2 --config 2
run_trace config3 --config 3
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/default/pmc_$c -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/default/pmc_$c.err
done
python tools/pmc_traffic.py $OUT/default/pmc_FETCH_SIZE $OUT/default/pmc_WRITE_SIZE $OUT/default/pmc_traffic.txt $OUT/default/pmc_traffic.json 3 "profiles/${R}_default/pmc_traffic.txt" > /dev/null
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/default/pmc_mfma -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/default/pmc_mfma.err
python tools/pmc_summary.py $OUT/default/pmc_mfma > $OUT/default/pmc_mfma.txt 2>&1
rm -rf $OUT/default/pmc_FETCH_SIZE $OUT/default/pmc_WRITE_SIZE $OUT/default/pmc_mfma
# the Hamming scan alone, three counter passes per code size (tools/hm_pmc.sh)
bash tools/hm_pmc.sh ${R}_256 256 > /dev/null 2>&1; cp gpurun_out/hm_pmc_${R}_256/summary.txt $OUT/default/pmc_hamming_256bit.txt
bash tools/hm_pmc.sh ${R}_512 512 > /dev/null 2>&1; cp gpurun_out/hm_pmc_${R}_512/summary.txt $OUT/default/pmc_hamming_512bit.txt
ls -R $OUT | head -40
