#!/bin/bash
# A/B of DRA_VAR_GATHER_ON_UPDATE (29183 = 12799 | 16384) against the default pipeline on one box.
TAG=${1:-r02y}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_agents.py -q -m gpu -x -p no:cacheprovider -k "async_pipeline or schedule_oracle" > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log | cut -c1-200
b() { name=$1; shift; env "$@" timeout 240 python bench.py --no-cpu-baseline --no-long-run ${ARGS} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json")); print("$name", round(d["value"], 1), d.get("parity_check", {}).get("ok"), d.get("host"))
except Exception as e:
    print("$name unreadable", e)
PY
}
ARGS="--variant 12799" b base A=1
ARGS="--variant 29183" b gou A=1
ARGS="--variant 29183" b gou_acu48 DRA_ACTOR_CUS=48
ARGS="--variant 29183" b gou_acu56 DRA_ACTOR_CUS=56
ARGS="--variant 12799" b base2 A=1
ARGS="--variant 29183" b gou2 A=1
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
timeout 200 python tools/phase_trace.py --variant 29183 > $OUT/phase_gou.json 2> $OUT/phase_gou.err; tail -2 $OUT/phase_gou.err | grep -v amdgpu.ids
python tools/phase_summary.py $OUT/phase_gou.json | cut -c1-150
unset DEEPRL_AMD_LIB
echo "== done"
