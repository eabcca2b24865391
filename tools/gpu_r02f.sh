#!/bin/bash
TAG=${1:-r02f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== gpu tests"; timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
for cus in 96 64 56; do
echo "== bench acus $cus"; DRA_ACTOR_CUS=$cus timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_acus$cus.json 2> $OUT/bench_acus$cus.err; head -c 170 $OUT/bench_acus$cus.json; echo; tail -3 $OUT/bench_acus$cus.err | grep -v amdgpu
done
echo "== done"
