#!/bin/bash
# rocprofv3 kernel statistics of the benchmarked pipeline for several DRA_TUNING masks (A/B of a variant bit under the real
# two-stream load).  usage: gpurun -- 'bash tools/gpu_prof_tuning.sh <tag> <mask> <mask> ...'
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
for M in "$@"; do
  (cd /tmp && DRA_TUNING=$M timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$M -- python $R/tools/ab_env.py --worker --steps ${STEPS:-1000} > $R/$OUT/run_$M.json 2> $R/$OUT/run_$M.err)
  python tools/prof_summary.py $OUT/prof_$M > $OUT/kernel_stats_$M.txt 2>&1
  python tools/prof_timeline.py $OUT/prof_$M 3000 1 > $OUT/timeline_$M.txt 2>&1
  rm -rf $OUT/prof_$M
  echo "== DRA_TUNING=$M"; cut -c1-90,100-160 $OUT/kernel_stats_$M.txt | head -18
done
