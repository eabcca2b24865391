#!/bin/bash
# rocprofv3 kernel statistics of the benchmarked pipeline for two builds of the library (A/B of one kernel's duration under the
# real two-stream load).  usage: gpurun -- 'bash tools/gpu_prof_ab.sh <tag> <libA.so> <libB.so> [steps]'
TAG=$1; A=$2; B=$3; STEPS=${4:-1200}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
for L in $A $B; do
  N=$(basename $L .so)
  (cd /tmp && DEEPRL_AMD_LIB=$R/$L timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$N -- python $R/tools/ab_env.py --worker --steps $STEPS > $R/$OUT/run_$N.json 2> $R/$OUT/run_$N.err)
  python tools/prof_summary.py $OUT/prof_$N > $OUT/kernel_stats_$N.txt 2>&1
  python tools/prof_timeline.py $OUT/prof_$N 3000 1 > $OUT/timeline_$N.txt 2>&1
  rm -rf $OUT/prof_$N
  echo "== $N"; cut -c1-80,100-160 $OUT/kernel_stats_$N.txt | head -20
done
