#!/bin/bash
# round 3, call a: parity of the new weight-gradient / late-fold kernels, then a same-box A/B of the masks
export TMPDIR=/tmp
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x -k "conv_bwd_fused or fc_bwd_fused or optim" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_agents.py -q -m gpu -p no:cacheprovider -k "fused_learner_matches_oracle or schedule_oracle" 2>&1 | tail -15
timeout 500 python tools/ab_variants.py --masks 193023,455167,979455,717311 --rounds 3 --steps 1500 2>gpurun_out/r03a/ab.err | cut -c1-1200 | tee gpurun_out/r03a/ab.jsonl
tail -5 gpurun_out/r03a/ab.err
