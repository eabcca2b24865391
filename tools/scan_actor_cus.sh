#!/bin/bash
# async step rate vs the number of CUs reserved for the actor chain (DRA_VAR_CU_PARTITION); same box, same process layout
for N in ${SCAN:-48 56 64 72 80}; do
  DRA_ACTOR_CUS=$N python tools/ab_variants.py --masks ${MASK:-4607} --rounds 2 --steps 1000 2>/dev/null | grep '"async"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('actor_cus', $N, 'updates_per_s', round(d['updates_per_s_median'], 1))"
done
