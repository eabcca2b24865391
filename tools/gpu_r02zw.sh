#!/bin/bash
# One gpurun call: where the fc4 K split changes a run (diagnosis), on-policy agents with the one-pass conv weight gradients at
# their batch sizes (DRA_ONESHOT_WGRAD_MAX_BATCH), QR-DQN after the loss-loop unroll, rocprofv3 of ppo_pixel.
TAG=${1:-r02zw}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== K split diagnosis"
DRA_FC4_KS=8 timeout 120 python tests/_switch_probe.py dqn $OUT/ks8.npz 2> $OUT/ks8.err
DRA_FC4_KS=14 timeout 120 python tests/_switch_probe.py dqn $OUT/ks14.npz 2> $OUT/ks14.err
DRA_FC4_KS=28 timeout 120 python tests/_switch_probe.py dqn $OUT/ks28.npz 2> $OUT/ks28.err
python tools/diag_ks.py $OUT/ks8.npz $OUT/ks14.npz; python tools/diag_ks.py $OUT/ks8.npz $OUT/ks28.npz | head -2
echo "== switch tests"
timeout 300 python -m pytest tests/test_gpu_env_switches.py tests/test_gpu_kernels.py -q -m gpu -x -p no:cacheprovider -k "switch or k_split or qr_loss or c51_loss or conv" 2>&1 | tail -3 | cut -c1-200
echo "== on-policy agents: one-pass weight gradients up to batch 64 (default) / 256 / 1024"
for mb in 64 256 1024; do
  DRA_ONESHOT_WGRAD_MAX_BATCH=$mb timeout 200 python tools/bench_agents.py --seconds 3 --cases a2c_pixel_16,ppo_pixel_8 > $OUT/onpolicy_mb$mb.jsonl 2> $OUT/onpolicy_mb$mb.err
  echo "(max batch $mb)"; cut -c1-200 $OUT/onpolicy_mb$mb.jsonl; tail -2 $OUT/onpolicy_mb$mb.err | cut -c1-200
done
echo "== qr / c51"
timeout 200 python tools/bench_agents.py --seconds 3 --cases qr_dqn_pixel_uniform_device,c51_pixel_uniform_device > $OUT/bench_agents.jsonl 2> $OUT/bench_agents.err; cut -c1-200 $OUT/bench_agents.jsonl
echo "== rocprofv3 ppo_pixel_8 (max batch 256)"
(cd /tmp && DRA_ONESHOT_WGRAD_MAX_BATCH=256 timeout 150 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_ppo -- python $R/tools/bench_agents.py --seconds 2 --cases ppo_pixel_8 > $R/$OUT/prof_ppo.log 2>&1)
grep '"case"' $OUT/prof_ppo.log | cut -c1-200
python tools/prof_summary.py $OUT/prof_ppo > $OUT/kernel_stats_ppo_pixel_8.txt 2>&1; head -34 $OUT/kernel_stats_ppo_pixel_8.txt | cut -c1-170
rm -rf $OUT/prof_ppo
echo "== done"
