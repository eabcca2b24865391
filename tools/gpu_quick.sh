#!/bin/bash
# A short gpurun call: (optionally) the GPU suite, then an environment A/B on the benchmarked pipeline.
# usage: gpurun -- 'bash tools/gpu_quick.sh <tag> <pytest -k expr or "all" or "none"> SETTING...'
TAG=$1; KEXPR=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rm -f gpurun_out/parity_errors.jsonl
if [ "$KEXPR" = "all" ]; then
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
elif [ "$KEXPR" != "none" ]; then
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -k "$KEXPR" > $OUT/pytest_gpu.log 2>&1
fi
[ -f $OUT/pytest_gpu.log ] && { grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -30; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1; tail -5 $OUT/pytest_gpu.log > $OUT/pytest_gpu_tail.txt; }
python tools/parity_summary.py gpurun_out/parity_errors.jsonl > $OUT/parity_errors.json 2>/dev/null
if [ $# -gt 0 ]; then
  timeout 900 python tools/ab_env.py --rounds ${AB_ROUNDS:-3} --steps ${AB_STEPS:-3000} "$@" > $OUT/ab_env.jsonl 2> $OUT/ab_env.err
  cat $OUT/ab_env.jsonl | cut -c1-900
fi
