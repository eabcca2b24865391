#!/bin/bash
# round 3, call c: the actor's env step as one launch (ACTOR_MEGA), the late-fold optimizer launch with one round of workgroups
export TMPDIR=/tmp
mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests/test_gpu_agents.py -q -m gpu -p no:cacheprovider -x -k "fused_learner_matches_oracle or schedule_oracle or fused_step_async_pipeline" 2>&1 | tail -8
# masks: round-2 default | + LATE_FOLD | + ACTOR_MEGA | + both | + WGRAD_ACC too
timeout 400 python tools/ab_variants.py --masks 193023,717311,1241599,1765887,2028031 --rounds 3 --steps 1500 2>>gpurun_out/r03c/ab.err | cut -c1-1200 | tee -a gpurun_out/r03c/ab.jsonl
for v in 1241599 1765887; do
  DEEPRL_AMD_LIB=deeprl_amd/lib/libdeeprl_amd_trace.so timeout 200 python tools/phase_trace.py --variant $v > gpurun_out/r03c/phase_async_$v.json 2>>gpurun_out/r03c/phase.err
  python tools/phase_summary.py gpurun_out/r03c/phase_async_$v.json 2>/dev/null | cut -c1-400 | head -40
done
tail -n 3 gpurun_out/r03c/ab.err gpurun_out/r03c/phase.err
