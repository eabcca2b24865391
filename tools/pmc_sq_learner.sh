#!/bin/bash
# SQ counter passes (8 SQ counters per pass, kernel-trace only) over the kernels of the DQN learner AT BATCH 32, inside
# the learner (tools/pmc_workload.py: in-order agent steps of BASELINE configs[1]).
# usage: gpurun -- 'bash tools/pmc_sq_learner.sh <tag>'   -> gpurun_out/<tag>/sq_learner_b32.json
TAG=${1:-pmc_sq}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
timeout 280 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$OUT/sq1 -- python $R/tools/pmc_workload.py --steps 30 --variant ${PMC_VARIANT:-255} --ring 100000 --no-calibration > $R/$OUT/sq1.log 2>&1
timeout 280 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/$OUT/sq2 -- python $R/tools/pmc_workload.py --steps 30 --variant ${PMC_VARIANT:-255} --ring 100000 --no-calibration > $R/$OUT/sq2.log 2>&1
cd $R
python tools/pmc_sq_summary.py $OUT/sq1 $OUT/sq2 > $OUT/sq_learner_b32.json 2> $OUT/sq_summary.err
head -c 2500 $OUT/sq_learner_b32.json; tail -2 $OUT/sq1.log; tail -2 $OUT/sq2.log; tail -3 $OUT/sq_summary.err
find $OUT/sq1 $OUT/sq2 -name "*kernel_trace*" -size +5M -delete
find $OUT/sq1 $OUT/sq2 -name "*.db" -size +5M -delete
