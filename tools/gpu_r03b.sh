#!/bin/bash
# round 3, call b: late-fold optimizer launch (valued arrival slots), per-layer choice of the accumulating weight gradient
export TMPDIR=/tmp
mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_agents.py -q -m gpu -p no:cacheprovider -k "fused_learner_matches_oracle or schedule_oracle" 2>&1 | tail -8
# masks: round-2 default | + LATE_FOLD | + WGRAD_ACC + LATE_FOLD  (ACC layers from the environment)
for layers in 7 4 6; do
  echo "== DRA_WGRAD_ACC_LAYERS=$layers"
  DRA_WGRAD_ACC_LAYERS=$layers timeout 300 python tools/ab_variants.py --masks 193023,717311,979455 --rounds 3 --steps 1500 2>>gpurun_out/r03b/ab.err | cut -c1-1200 | tee -a gpurun_out/r03b/ab_layers$layers.jsonl
done
for v in 193023 717311; do
  DEEPRL_AMD_LIB=deeprl_amd/lib/libdeeprl_amd_trace.so timeout 200 python tools/phase_trace.py --variant $v > gpurun_out/r03b/phase_async_$v.json 2>>gpurun_out/r03b/phase.err
  python tools/phase_summary.py gpurun_out/r03b/phase_async_$v.json 2>/dev/null | head -40
done
tail -3 gpurun_out/r03b/ab.err gpurun_out/r03b/phase.err
