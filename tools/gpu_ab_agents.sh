#!/bin/bash
# same-box A/B of environment toggles on tools/bench_agents.py cases: gpu_ab_agents.sh <tag> <cases> VAR=a VAR=b ...
TAG=$1; CASES=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
for kv in "$@"; do
  echo "== $kv (rep $rep)"
  env $kv python tools/bench_agents.py --seconds 3 --cases $CASES 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    d = json.loads(line); print('   ', d.get('case'), d.get('updates_per_s', d.get('error')))
" | tee -a $OUT/ab_$kv.txt
done
done
