#!/bin/bash
# The whole GPU suite, then the driver's own command under two DRA_TUNING masks, interleaved (A/B of a default-mask change as the
# driver will see it).  usage: gpurun -- 'bash tools/gpu_suite_and_driver_ab.sh <tag> <maskA> <maskB>'
TAG=$1; A=$2; B=$3; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rm -f gpurun_out/parity_errors.jsonl
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -30; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1
tail -5 $OUT/pytest_gpu.log > $OUT/pytest_gpu_tail.txt
for i in 1 2 3; do for M in $A $B; do
  DRA_TUNING=$M timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(json.dumps({'mask': $M, 'driver_form': d['value'], 'long_run': d['long_run']['value'], 'variant': d['config']['kernel_variant'], 'parity_ok': d['parity_check'].get('ok')}))" | tee -a $OUT/driver_ab.jsonl
done; done
