#!/bin/bash
TAG=${1:-r02ze}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_agents.py -q -m gpu -x -p no:cacheprovider -k "schedule_oracle or async_pipeline" > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log | cut -c1-200
for rep in 1 2; do for v in 29183 61951; do
  python bench.py --no-cpu-baseline --no-long-run --variant $v > $OUT/bench_v${v}_$rep.json 2> $OUT/bench_v${v}_$rep.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_v${v}_$rep.json")); print("variant $v rep $rep:", round(d["value"], 1), "parity", d.get("parity_check", {}).get("ok"), d.get("parity_check", {}).get("max_abs_param_err"))
except Exception as e:
    print("variant $v rep $rep: unreadable", e); print(open("$OUT/bench_v${v}_$rep.err").read()[-600:])
PY
done; done
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
python tools/phase_trace.py --variant 61951 > $OUT/phase_rd.json 2>/dev/null
python tools/phase_summary.py $OUT/phase_rd.json | cut -c1-200
