#!/bin/bash
TAG=${1:-r02u}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== gpu tests"; timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log | cut -c1-200
echo "== bench"; timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d=json.load(open("$OUT/bench.json")); print({k:d[k] for k in ("value","ms_per_step","parity_check","roofline","cpu_baseline","agent_api","kernel_ms","update_kernel_ms_sum") if k in d})
PY
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
timeout 200 python tools/phase_trace.py > $OUT/phase_async.json 2> $OUT/phase.err; python tools/phase_summary.py $OUT/phase_async.json | cut -c1-260
