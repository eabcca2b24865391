#!/bin/bash
# One gpurun call of a round: the full GPU suite, bench (default + the driver's command), rocprofv3 kernel stats + timeline of
# the same command, the two --pmc traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only), SQ counters at
# batch 32, phase traces (update / actor chains, the large-batch conv forward), NatureConv forward + backward at batch
# 256 / 512 / 1024 with counters, the gather microbench, the PER timeline, agent benches.
# Everything under gpurun_out/<tag>/ (copy what should be judged into profiles/).   usage: gpurun -- 'bash tools/gpu_round.sh r04x'
TAG=${1:-r04x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PV=${PMC_VARIANT:-524543}     # in-order learner: 255 + LATE_FOLD
nproc > $OUT/nproc.txt
rm -f gpurun_out/parity_errors.jsonl
echo "== GPU tests"; timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -40; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; tail -5 $OUT/pytest_gpu.log > $OUT/pytest_gpu_tail.txt
python tools/parity_summary.py gpurun_out/parity_errors.jsonl > $OUT/parity_errors.json 2>/dev/null; head -c 600 $OUT/parity_errors.json; echo
echo "== bench" ; timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 400 $OUT/bench.json; echo
echo "== bench (driver command)" ; timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; head -c 300 $OUT/bench_driver_cmd.json; echo
echo "== rocprofv3 kernel trace" ; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -- python $R/bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-parity-check > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_bench.err)
python tools/prof_summary.py $OUT/prof > $OUT/rocprofv3_kernel_stats.txt 2>&1; head -32 $OUT/rocprofv3_kernel_stats.txt
python tools/prof_timeline.py $OUT/prof 200 2 > $OUT/timeline.txt 2>&1
find $OUT/prof -name "*.db" -size +20M -delete
echo "== pmc passes"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python $R/tools/pmc_workload.py --steps 40 --variant $PV > $R/$OUT/pmc_fetch.log 2>&1); tail -1 $OUT/pmc_fetch.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python $R/tools/pmc_workload.py --steps 40 --variant $PV > $R/$OUT/pmc_write.log 2>&1); tail -1 $OUT/pmc_write.log
# the chained launches run only in the two-stream pipeline: a second pair of passes there, WITHOUT the variants whose launches wait
# for each other across streams on device words (counter collection serialises kernels): default - FLAG_SYNC - LANE_EAGER - ACTOR_PERSIST
PV2=${PMC_VARIANT_ASYNC:-110817791}
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch/async -- python $R/tools/pmc_workload.py --steps 40 --variant $PV2 --async-actor --no-calibration > $R/$OUT/pmc_fetch_async.log 2>&1); tail -1 $OUT/pmc_fetch_async.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write/async -- python $R/tools/pmc_workload.py --steps 40 --variant $PV2 --async-actor --no-calibration > $R/$OUT/pmc_write_async.log 2>&1); tail -1 $OUT/pmc_write_async.log
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err; head -c 1200 $OUT/pmc_traffic.json; tail -2 $OUT/pmc_traffic.err
find $OUT/pmc_fetch $OUT/pmc_write -name "*.db" -size +20M -delete
find $OUT/pmc_fetch $OUT/pmc_write -name "*kernel_trace*" -size +20M -delete
echo "== SQ counters at batch 32"; PMC_VARIANT=$PV bash tools/pmc_sq_learner.sh $TAG > $OUT/sq.log 2>&1; tail -3 $OUT/sq.log
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
echo "== phase traces"; timeout 200 python tools/phase_trace.py > $OUT/phase_async.json 2> $OUT/phase.err; python tools/phase_summary.py $OUT/phase_async.json | grep -E "chain|env step"
timeout 200 python tools/phase_trace.py --sync > $OUT/phase_sync.json 2>> $OUT/phase.err; python tools/phase_summary.py $OUT/phase_sync.json | grep -E "chain|env step"
echo "== NatureConv forward at batch 1024: phases"; timeout 120 python tools/phase_conv_big.py 1024 > $OUT/phase_conv_big_b1024.json 2>> $OUT/phase.err
unset DEEPRL_AMD_LIB
echo "== NatureConv forward + backward at batch 256 / 512 / 1024, gather microbench"; bash tools/conv_big_counters.sh $TAG > $OUT/conv_big_stdout.txt 2>&1; tail -5 $OUT/conv_big_stdout.txt | cut -c1-300
python tools/bench_kernels.py > $OUT/bench_kernels.json 2> $OUT/bench_kernels.err
echo "== prioritized agent step: kernel timeline"; (cd /tmp && timeout 200 rocprofv3 --kernel-trace -d $R/$OUT/prof_per -- python $R/tools/bench_agents.py --seconds 2 --cases dqn_pixel_per_device > $R/$OUT/prof_per.log 2>&1); python tools/prof_timeline.py $OUT/prof_per 3000 1 > $OUT/timeline_per.txt 2>&1; rm -rf $OUT/prof_per; grep -c . $OUT/timeline_per.txt
echo "== agents bench"; timeout 400 python tools/bench_agents.py > $OUT/bench_agents.jsonl 2> $OUT/bench_agents.err; cat $OUT/bench_agents.jsonl | cut -c1-300
echo "== launch contract: 2 ranks on this box (gloo barrier, replicas share the GPU)"; timeout 300 python bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline --no-parity-check > $OUT/bench_2rank_one_box.json 2> $OUT/bench_2rank_one_box.err; head -c 300 $OUT/bench_2rank_one_box.json; echo
echo "== done"
