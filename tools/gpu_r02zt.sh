#!/bin/bash
# One gpurun call (round 2, late): the cooperative clip + optimizer launch (DRA_VAR_COOP_OPT), prefetched minibatch indices
# (DRA_VAR_IDX_PREFETCH), 14-way K split of the update's fc4 forward, LDS-staged actor fc4 -- parity tests of the new
# kernels, same-box A/B of bench.py (2 repetitions, interleaved), full GPU test suite, phase traces.
TAG=${1:-r02zt}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NEW="DRA_TUNING=258559 DRA_FC4_KS=14 DRA_ACTOR_FC4_LDS=1"
echo "== new-kernel tests"
timeout 300 python -m pytest tests -q -m gpu -x -p no:cacheprovider \
  -k "clip_step_coop or sqnorm or fused_learner_matches_oracle or fused_step_async_pipeline or schedule_oracle" > $OUT/pytest_new.log 2>&1
tail -4 $OUT/pytest_new.log | cut -c1-300
echo "== A/B (updates/s, parity)"
i=0
for rep in 1 2; do
  for cfg in "base|DRA_TUNING=61951" "coop|DRA_TUNING=127487" "prefetch|DRA_TUNING=193023" "coop+prefetch|DRA_TUNING=258559" \
             "ks14|DRA_TUNING=61951 DRA_FC4_KS=14" "afc4lds|DRA_TUNING=61951 DRA_ACTOR_FC4_LDS=1" "all|$NEW"; do
    name=${cfg%%|*}; kv=${cfg#*|}
    env $kv timeout 90 python bench.py --no-cpu-baseline --no-long-run > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${name}_$rep.json"))
    print("%-14s rep $rep: %8.1f updates/s  parity %s  coop %s  dom %s %.2f us" % ("$name", d["value"], d.get("parity_check", {}).get("ok"),
          d.get("coop_optimizer", {}).get("in_use"), d["roofline"]["kernel"], 1e3 * d["roofline"]["avg_ms"]))
except Exception as e:
    print("$name rep $rep: unreadable", e)
PY
  done
done
echo "== tests under the new configuration"
env $NEW timeout 300 python -m pytest tests -q -m gpu -x -p no:cacheprovider \
  -k "schedule_oracle or fast_path or pixel_agents or launcher or fused_step_async_pipeline" > $OUT/pytest_newcfg.log 2>&1
tail -4 $OUT/pytest_newcfg.log | cut -c1-300
echo "== full GPU suite (default configuration)"
timeout 420 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
tail -4 $OUT/pytest_gpu.log | cut -c1-300
echo "== phase traces"
export DEEPRL_AMD_LIB=$GRAFT_REPO_ROOT/deeprl_amd/lib/libdeeprl_amd_trace.so
timeout 90 python tools/phase_trace.py > $OUT/phase_async_base.json 2> $OUT/phase_base.err; python tools/phase_summary.py $OUT/phase_async_base.json | cut -c1-260
env $NEW timeout 90 python tools/phase_trace.py > $OUT/phase_async_new.json 2> $OUT/phase_new.err; python tools/phase_summary.py $OUT/phase_async_new.json | cut -c1-260
unset DEEPRL_AMD_LIB
echo "== done"
