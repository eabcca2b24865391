#!/bin/bash
# Round-5 evidence in one gpurun call (≈ 6 min): rocprofv3 kernel stats of bench.py (headline) and of the ppo_continuous workload,
# the rollout-batch conv forward / backward with rocprofv3 + counters, the ppo_mlp phase profile, the agent lines.
# usage: gpurun -- 'bash tools/gpu_evidence.sh r05x'       (copy what should be judged into profiles/)
TAG=${1:-r05x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== bench"; timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 300 $OUT/bench.json; echo
echo "== bench (driver command)"; timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; head -c 200 $OUT/bench_driver_cmd.json; echo
echo "== bench ppo_continuous"; timeout 300 python bench.py --workload ppo_continuous --steps 20 --warmup 5 > $OUT/bench_ppo_continuous.json 2> $OUT/bench_ppo.err; head -c 300 $OUT/bench_ppo_continuous.json; echo
echo "== rocprofv3: bench.py"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -- python $R/bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-parity-check > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_bench.err)
python tools/prof_summary.py $OUT/prof > $OUT/rocprofv3_kernel_stats.txt 2>&1; head -16 $OUT/rocprofv3_kernel_stats.txt | cut -c1-150
find $OUT/prof -name "*.db" -size +20M -delete
echo "== rocprofv3: ppo_continuous"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_ppo -- python $R/bench.py --workload ppo_continuous --steps 20 --warmup 5 --no-cpu-baseline > $R/$OUT/prof_ppo.json 2> $R/$OUT/prof_ppo.err)
python tools/prof_summary.py $OUT/prof_ppo > $OUT/rocprofv3_kernel_stats_ppo_continuous.txt 2>&1; head -8 $OUT/rocprofv3_kernel_stats_ppo_continuous.txt | cut -c1-150
rm -rf $OUT/prof_ppo
echo "== ppo_mlp phases"; timeout 200 python tools/prof_ppo_mlp.py 2> /dev/null | grep -v amdgpu.ids > $OUT/prof_ppo_mlp.json; tail -16 $OUT/prof_ppo_mlp.json
echo "== conv at rollout batch sizes"; bash tools/conv_big_counters.sh $TAG > $OUT/conv_big_stdout.txt 2>&1; python - <<PY
import json
for line in open("$OUT/conv_big.jsonl"):
    d = json.loads(line)
    print(d["batch"], {k: (v.get("rocprofv3_us"), v.get("rocprofv3_frac"), v.get("mfma_util_counter")) for k, v in d["kernels"].items()})
PY
for b in 256 512 1024; do python tools/conv_big_bwd.py $b 30 roles 2>/dev/null | grep batch >> $OUT/conv_big_roles.jsonl; done
echo "== agents"; timeout 400 python tools/bench_agents.py --seconds 4 > $OUT/bench_agents.jsonl 2> $OUT/bench_agents.err; cut -c1-220 $OUT/bench_agents.jsonl
echo "== done"
