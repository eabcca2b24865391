#!/bin/bash
TAG=${1:-r02d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in 0 1 2 3 4 7; do
echo "== bench nosync $v"; DRA_DEBUG_NOSYNC=$v timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_ns$v.json 2> $OUT/bench_ns$v.err; head -c 170 $OUT/bench_ns$v.json; echo
done
echo "== bench nosync 7 acus 64"; DRA_ACTOR_CUS=64 DRA_DEBUG_NOSYNC=7 timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_ns7_acu64.json 2> $OUT/bench_ns7.err; head -c 170 $OUT/bench_ns7_acu64.json; echo
echo "== done"
