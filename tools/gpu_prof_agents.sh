#!/bin/bash
# rocprofv3 kernel stats of tools/bench_agents.py cases: gpu_prof_agents.sh <tag> <case>[,<case>...]
TAG=$1; CASES=$2; R=$GRAFT_REPO_ROOT; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for c in ${CASES//,/ }; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$c -- python $R/tools/bench_agents.py --seconds 2 --cases $c > $R/$OUT/prof_$c.log 2>&1)
  echo "== $c"; grep '"case"' $OUT/prof_$c.log | cut -c1-200
  python tools/prof_summary.py $OUT/prof_$c > $OUT/kernel_stats_$c.txt 2>&1; head -34 $OUT/kernel_stats_$c.txt | cut -c1-170
  rm -rf $OUT/prof_$c
done
