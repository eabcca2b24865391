#!/bin/bash
# round-2 call A: baseline on this box + phase traces + SQ counters at batch 32 + env-var A/Bs.
TAG=${1:-r02a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
nproc > $OUT/nproc.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench default"; timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 600 $OUT/bench_default.json; echo
echo "== bench persist conv fwd at B=32"; DRA_CONV_PT_BATCH=32 timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_persist32.json 2> $OUT/bench_persist32.err; head -c 400 $OUT/bench_persist32.json; echo
for cus in 64 128; do
  echo "== bench actor cus $cus"; DRA_ACTOR_CUS=$cus timeout 200 python bench.py --no-cpu-baseline --steps 1500 > $OUT/bench_acus$cus.json 2> $OUT/bench_acus$cus.err; head -c 200 $OUT/bench_acus$cus.json; echo
done
echo "== bench sync actor"; timeout 200 python bench.py --no-cpu-baseline --sync-actor --steps 1500 > $OUT/bench_sync.json 2>&1; head -c 200 $OUT/bench_sync.json; echo
echo "== bench no actor"; timeout 200 python bench.py --no-cpu-baseline --no-actor --steps 1500 > $OUT/bench_noactor.json 2>&1; head -c 200 $OUT/bench_noactor.json; echo
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
echo "== phase trace async"; timeout 200 python tools/phase_trace.py > $OUT/phase_async.json 2> $OUT/phase_async.err; tail -3 $OUT/phase_async.err
echo "== phase trace sync";  timeout 200 python tools/phase_trace.py --sync > $OUT/phase_sync.json 2> $OUT/phase_sync.err; tail -3 $OUT/phase_sync.err
echo "== phase trace async persist32"; DRA_CONV_PT_BATCH=32 timeout 200 python tools/phase_trace.py > $OUT/phase_async_persist32.json 2> $OUT/phase_async_persist32.err; tail -3 $OUT/phase_async_persist32.err
unset DEEPRL_AMD_LIB
python - <<PY
import json
for f in ("phase_async", "phase_sync"):
    try:
        d = json.load(open("$OUT/%s.json" % f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print("==", f)
    for k, v in d["kernels"].items():
        print("%-16s wgs %5d start %7.2f span %6.2f wg_mean %5.2f  starts %s  phases %s" % (k, v["workgroups"], v["t_start_us"], v["span_us"], v["wg_dur_us"]["mean"], v["wg_start_us_hist(0,1,2,4,6,8,12,16+)"], {a.split()[0]: b["mean"] for a, b in v["phases_us"].items()}))
PY
echo "== SQ counters"; bash tools/pmc_sq_learner.sh $TAG > $OUT/sq.log 2>&1; tail -5 $OUT/sq.log
echo "== done"
