#!/bin/bash
# NatureConv forward / backward at batch 256 / 512 / 1024: HIP-event rates (tools/conv_big_bwd.py), the same
# command under `rocprofv3 --kernel-trace` (per-kernel durations) and under a PMC pass (SQ_VALU_MFMA_BUSY_CYCLES,
# GRBM_GUI_ACTIVE, instruction mix; kernel-trace only, no other trace domain).  -> gpurun_out/<tag>/conv_big.jsonl
# usage: gpurun -- 'bash tools/conv_big_counters.sh <tag> [batches]'
TAG=${1:-r04conv}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
BATCHES=${2:-"256 512 1024"}
: > $OUT/conv_big.jsonl
for b in $BATCHES; do
  timeout 120 python tools/conv_big_bwd.py $b > $OUT/conv_big_events_b$b.json 2> $OUT/conv_big_b$b.err
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace -d $R/$OUT/cb_prof$b -- python $R/tools/conv_big_bwd.py $b 10 > /dev/null 2>> $R/$OUT/conv_big_b$b.err)
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/$OUT/cb_pmc$b -- python $R/tools/conv_big_bwd.py $b 4 > /dev/null 2>> $R/$OUT/conv_big_b$b.err)
  python tools/conv_big_summary.py $OUT/conv_big_events_b$b.json $OUT/cb_prof$b $OUT/cb_pmc$b >> $OUT/conv_big.jsonl 2>> $OUT/conv_big_b$b.err
  find $OUT/cb_prof$b $OUT/cb_pmc$b -name "*.db" -size +5M -delete
  find $OUT/cb_prof$b $OUT/cb_pmc$b -name "*kernel_trace*" -size +5M -delete
done
cat $OUT/conv_big.jsonl | cut -c1-1500
