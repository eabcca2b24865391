#!/bin/bash
TAG=${1:-diag}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for args in "async -1 ortho" "async -1 normal"; do
  echo "== $args"; timeout 300 python tests/diag_schedule.py $args 2>&1 | grep -v amdgpu.ids | tee -a $OUT/diag.log | cut -c1-220
done
