#!/bin/bash
# BASELINE config 5 (one rank's shard of the 100 M-item / 10^9-rating graph) profile evidence on the GPU box:
#   gpurun_out/profiles_$1_config5/{bench.json, kernel_stats.csv, bench_under_rocprof.json, pmc_traffic.txt, pmc_traffic.json,
#   pmc_sampler.txt, pmc_calibration.txt}
# usage: tools/profile_config5.sh r03   (through gpurun; rocprofv3 gets the program itself after --)
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_${R}_config5
mkdir -p $OUT
python bench.py --config 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; python tools/show_bench.py $OUT/bench.json | head -14
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o t -- python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $OUT/raw/t_kernel_stats.csv $OUT/kernel_stats.csv; rm -rf $OUT/raw
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$c.err
  echo "pmc $c rc=$?"
done
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_traffic.txt $OUT/pmc_traffic.json 2 "profiles/${R}_config5/pmc_traffic.txt, bench.py --config 5" > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_sq -o p -- python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
python tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sampler.txt 2>&1
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts in the sampler's access shapes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/cal_$c -o p -- python tools/pmc_calibrate.py > $OUT/cal_$c.json 2> $OUT/cal_$c.err
done
python tools/pmc_calibrate_report.py $OUT/cal_FETCH_SIZE $OUT/cal_WRITE_SIZE $OUT/cal_FETCH_SIZE.json > $OUT/pmc_calibration.txt 2>&1
cat $OUT/pmc_calibration.txt
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq $OUT/cal_FETCH_SIZE $OUT/cal_WRITE_SIZE
ls $OUT
