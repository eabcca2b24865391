#!/bin/bash
# usage: gpurun -- 'bash tools/gpu_quick_tests.sh <tag> [pytest -k expression]'
TAG=${1:-quick}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$2" ]; then timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "$2" > $OUT/pytest_gpu.log 2>&1
else timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; fi
tail -25 $OUT/pytest_gpu.log | cut -c1-250
