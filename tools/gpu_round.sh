#!/bin/bash
# One gpurun call: GPU parity tests, bench, rocprofv3 kernel stats, the two --pmc passes (FETCH_SIZE / WRITE_SIZE,
# separate runs, kernel-trace only), kernel micro-bench.  Everything under gpurun_out/<tag>/.
# usage: gpurun --timeout 400 -- 'timeout 380 bash tools/gpu_round.sh <tag>'   (rocprofv3 runs of bench.py end with a segfault inside the
# profiler's finalizer when CU-masked streams exist; the result files are complete -- the PMC passes use --variant 255)
TAG=${1:-r01c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
nproc > $OUT/nproc.txt
echo "== all gpu tests" ; timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log
echo "== bench" ; timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
echo "== rocprofv3 kernel trace" ; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -- python $R/bench.py --steps 600 --warmup 100 --no-cpu-baseline > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_bench.err)
python tools/prof_summary.py $OUT/prof > $OUT/rocprofv3_kernel_stats.txt 2>&1; head -30 $OUT/rocprofv3_kernel_stats.txt
python tools/prof_timeline.py $OUT/prof 200 2 > $OUT/timeline.txt 2>&1
find $OUT/prof -name "*.db" -size +20M -delete
echo "== pmc FETCH_SIZE" ; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python $R/tools/pmc_workload.py --steps 40 --variant 255 > $R/$OUT/pmc_fetch.log 2>&1); tail -2 $OUT/pmc_fetch.log
echo "== pmc WRITE_SIZE" ; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python $R/tools/pmc_workload.py --steps 40 --variant 255 > $R/$OUT/pmc_write.log 2>&1); tail -2 $OUT/pmc_write.log
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err; head -c 1500 $OUT/pmc_traffic.json; tail -3 $OUT/pmc_traffic.err
find $OUT/pmc_fetch $OUT/pmc_write -name "*.db" -size +20M -delete
find $OUT/pmc_fetch $OUT/pmc_write -name "*kernel_trace*" -size +20M -delete
echo "== kernel microbench" ; timeout 300 python tools/bench_kernels.py > $OUT/kernel_microbench.json 2> $OUT/kernel_microbench.err; cat $OUT/kernel_microbench.json | head -c 3000
echo "== done"
