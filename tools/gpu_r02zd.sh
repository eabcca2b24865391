#!/bin/bash
TAG=${1:-r02zd}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_agents.py -q -m gpu -x -p no:cacheprovider -k "schedule_oracle or async_pipeline" > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log | cut -c1-200
bash tools/gpu_ab_bench.sh $TAG DRA_ACTOR_KSPLIT=1 DRA_ACTOR_KSPLIT=0
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
for k in 1 0; do
  DRA_ACTOR_KSPLIT=$k python tools/phase_trace.py > $OUT/phase_k$k.json 2>/dev/null
  echo "== ksplit $k"; python tools/phase_summary.py $OUT/phase_k$k.json | grep -E "actor|chain|env step" | cut -c1-210
done
unset DEEPRL_AMD_LIB
echo "== ppo one-shot wgrad at batch 256"
bash tools/gpu_ab_agents.sh $TAG ppo_pixel_8 DRA_ONESHOT_WGRAD_MAX_BATCH=64 DRA_ONESHOT_WGRAD_MAX_BATCH=256
