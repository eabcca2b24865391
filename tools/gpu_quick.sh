#!/bin/bash
# usage: gpurun -- 'bash tools/gpu_quick.sh "<pytest -k expr>" <masks>'   quick parity subset + same-box A/B + step timeline
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "$1" 2>&1 | tail -3
timeout 400 python tools/ab_variants.py --masks ${2:-511} --rounds 3 --steps 1000 2>/dev/null | cut -c1-900
python tools/diag_trace.py --variant ${3:-511} --brief
