#!/bin/bash
# round 3, call d: actor [conv1][conv2][conv3+fc4] (ACTOR_MEGA mode 1), late-fold launch in one round of workgroups
export TMPDIR=/tmp
mkdir -p gpurun_out/r03d
timeout 900 python -m pytest tests/test_gpu_agents.py -q -m gpu -p no:cacheprovider -k "fused_learner_matches_oracle or schedule_oracle or fused_step_async_pipeline" 2>&1 | tail -12
timeout 400 python tools/ab_variants.py --masks 193023,717311,1241599,1765887 --rounds 3 --steps 1500 2>>gpurun_out/r03d/ab.err | cut -c1-260 | tee -a gpurun_out/r03d/ab.jsonl
for v in 1765887; do
  DEEPRL_AMD_LIB=deeprl_amd/lib/libdeeprl_amd_trace.so timeout 200 python tools/phase_trace.py --variant $v > gpurun_out/r03d/phase_async_$v.json 2>>gpurun_out/r03d/phase.err
  python tools/phase_summary.py gpurun_out/r03d/phase_async_$v.json 2>/dev/null | cut -c1-330 | head -40
done
tail -n 3 gpurun_out/r03d/ab.err gpurun_out/r03d/phase.err
