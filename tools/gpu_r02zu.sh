#!/bin/bash
# One gpurun call: A/B of the LDS-staged actor fc4 (now default), 14-way fc4 K split, async-copy index prefetch, actor CU
# counts; rocprofv3 kernel stats of the config-4 agents; phase trace of the candidate configuration; parity tests under it.
TAG=${1:-r02zu}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
NEW="DRA_TUNING=193023 DRA_FC4_KS=14"
echo "== A/B (updates/s, parity)"
for rep in 1 2; do
  for cfg in "base|DRA_TUNING=61951" "lds0|DRA_TUNING=61951 DRA_ACTOR_FC4_LDS=0" "ks14|DRA_TUNING=61951 DRA_FC4_KS=14" "prefetch|DRA_TUNING=193023" \
             "new|$NEW" "new_acu16|$NEW DRA_ACTOR_CUS=16" "new_acu24|$NEW DRA_ACTOR_CUS=24" "new_acu40|$NEW DRA_ACTOR_CUS=40"; do
    name=${cfg%%|*}; kv=${cfg#*|}
    env $kv timeout 90 python bench.py --no-cpu-baseline --no-long-run > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${name}_$rep.json"))
    k = d["kernel_ms"]
    print("%-10s rep $rep: %8.1f updates/s  parity %s  conv1_fwd %.2f fc4_fwd %.2f head %.2f" % ("$name", d["value"], d.get("parity_check", {}).get("ok"),
          1e3 * k["conv1_fwd"], 1e3 * k["fc4_fwd"], 1e3 * k["head_loss"]))
except Exception as e:
    print("$name rep $rep: unreadable", e)
PY
  done
done
echo "== tests under the candidate configuration"
env $NEW timeout 300 python -m pytest tests -q -m gpu -x -p no:cacheprovider \
  -k "schedule_oracle or fast_path or pixel_agents or launcher or fused_step_async_pipeline" > $OUT/pytest_newcfg.log 2>&1
tail -3 $OUT/pytest_newcfg.log | cut -c1-300
echo "== phase trace (candidate)"
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
env $NEW timeout 90 python tools/phase_trace.py > $OUT/phase_async_new.json 2> $OUT/phase_new.err; python tools/phase_summary.py $OUT/phase_async_new.json | cut -c1-260
unset DEEPRL_AMD_LIB
echo "== rocprofv3 kernel stats: config-4 agents"
for c in c51_pixel_uniform_device qr_dqn_pixel_uniform_device dqn_pixel_per_device; do
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$c -- python $R/tools/bench_agents.py --seconds 2 --cases $c > $R/$OUT/prof_$c.log 2>&1)
  echo "== $c"; grep '"case"' $OUT/prof_$c.log | cut -c1-200
  python tools/prof_summary.py $OUT/prof_$c > $OUT/kernel_stats_$c.txt 2>&1; head -40 $OUT/kernel_stats_$c.txt | cut -c1-170
  rm -rf $OUT/prof_$c
done
echo "== done"
