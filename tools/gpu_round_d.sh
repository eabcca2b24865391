#!/bin/bash
# full GPU tests + bench (default variant) + A/B against the previous defaults on the SAME box
TAG=${1:-r01d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-900 $OUT/bench.json
timeout 400 python tools/ab_variants.py --masks 127,255,511,1023 --rounds 3 --steps 1000 > $OUT/ab.jsonl 2> $OUT/ab.err; cut -c1-200 $OUT/ab.jsonl
python tools/diag_trace.py --variant 511 --brief; python tools/diag_trace.py --variant 255 --brief
