#!/bin/bash
TAG=${1:-r02k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "schedule_oracle or fused_learner_matches" > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log | cut -c1-300
echo "== bench default (driver command)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cat $OUT/bench_driver.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','short_run','long_run','parity_check','cpu_baseline','roofline') if k in d})"; tail -3 $OUT/bench_driver.err | grep -v amdgpu
echo "== bench 2-rank self spawn"; timeout 600 python bench.py --gpus 2 --steps 600 --warmup 100 --no-cpu-baseline > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; head -c 300 $OUT/bench_2rank.json; echo; tail -5 $OUT/bench_2rank.err | grep -v amdgpu
