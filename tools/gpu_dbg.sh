#!/bin/bash
OUT=gpurun_out/dbg; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 400 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "fast_path or schedule_oracle" > $OUT/pytest_$i.log 2>&1
echo "run $i: $(grep -E 'passed|failed' $OUT/pytest_$i.log | tail -1 | cut -c1-120) $(grep -c 'Fatal Python' $OUT/pytest_$i.log) fatal"
done
echo "== full suite"; timeout 600 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1; grep -E "^(FAILED|ERROR)|Fatal" $OUT/pytest_gpu.log | head -5 | cut -c1-200
echo "== c51 / qr: fused actor on / off"
for v in 1 0; do DRA_ACTOR_DIST_FUSED=$v timeout 200 python tools/bench_agents.py --seconds 3 --cases c51_pixel_uniform_device,qr_dqn_pixel_uniform_device 2>/dev/null | cut -c1-140; done
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_c51 -- python $R/tools/bench_agents.py --seconds 2 --cases c51_pixel_uniform_device > $R/$OUT/prof_c51.log 2>&1)
python tools/prof_summary.py $OUT/prof_c51 > $OUT/kernel_stats_c51.txt 2>&1; head -28 $OUT/kernel_stats_c51.txt | cut -c1-150; rm -rf $OUT/prof_c51
