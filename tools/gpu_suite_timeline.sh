#!/bin/bash
# The whole GPU suite, then the rocprofv3 timeline of the benchmarked pipeline under one DRA_TUNING mask.
# usage: gpurun -- 'bash tools/gpu_suite_timeline.sh <tag> <mask>'
TAG=$1; M=$2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rm -f gpurun_out/parity_errors.jsonl
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -30; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1
tail -5 $OUT/pytest_gpu.log > $OUT/pytest_gpu_tail.txt
bash tools/gpu_timeline.sh $TAG $M
