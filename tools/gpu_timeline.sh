#!/bin/bash
# rocprofv3 kernel trace of the benchmarked pipeline for one DRA_TUNING mask -> per-queue timeline of two steady-state steps
# usage: gpurun -- 'bash tools/gpu_timeline.sh <tag> <mask>'
TAG=$1; M=$2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
(cd /tmp && DRA_TUNING=$M timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$M -- python $R/tools/ab_env.py --worker --steps 1000 > $R/$OUT/run_$M.json 2> $R/$OUT/run_$M.err)
python tools/prof_summary.py $OUT/prof_$M > $OUT/kernel_stats_$M.txt 2>&1
python tools/prof_timeline.py $OUT/prof_$M 600 3 > $OUT/timeline_$M.txt 2>&1
rm -rf $OUT/prof_$M
cut -c1-150 $OUT/timeline_$M.txt | head -60
