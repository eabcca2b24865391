#!/bin/bash
# same-box A/B of environment toggles on bench.py: gpu_ab_bench.sh <tag> VAR=a VAR=b ...   (2 repetitions each, interleaved)
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
for kv in "$@"; do
  env $kv python bench.py --no-cpu-baseline --no-long-run > $OUT/bench_${kv}_$rep.json 2> $OUT/bench_${kv}_$rep.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${kv}_$rep.json")); print("$kv rep $rep:", round(d["value"], 1), "parity", d.get("parity_check", {}).get("ok"))
except Exception as e:
    print("$kv rep $rep: unreadable", e)
PY
done
done
