#!/bin/bash
TAG=${1:-r02zx}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for n in 41 43 48 60; do
  for ks in 8 14 28; do PROBE_STEPS=$n DRA_FC4_KS=$ks timeout 120 python tests/_switch_probe.py dqn $OUT/ks${ks}_$n.npz 2> $OUT/ks${ks}_$n.err; done
  echo "== $n steps: 8 vs 14"; python tools/diag_ks.py $OUT/ks8_$n.npz $OUT/ks14_$n.npz | grep -E "actions|fc4.weight|conv1.weight"
  echo "== $n steps: 8 vs 28"; python tools/diag_ks.py $OUT/ks8_$n.npz $OUT/ks28_$n.npz | grep -E "actions|fc4.weight|conv1.weight"
done
