#!/bin/bash
TAG=${1:-r02v}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== bench"; timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d=json.load(open("$OUT/bench.json")); print({k:d[k] for k in ("value","ms_per_step","host","agent_api","cpu_baseline") if k in d})
PY
tail -3 $OUT/bench.err | grep -v amdgpu
echo "== kernel microbench"; timeout 300 python tools/bench_kernels.py > $OUT/kernel_microbench.json 2> $OUT/kernel_microbench.err; head -c 2500 $OUT/kernel_microbench.json; tail -2 $OUT/kernel_microbench.err | grep -v amdgpu
