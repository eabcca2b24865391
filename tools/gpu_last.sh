#!/bin/bash
OUT=gpurun_out/last2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 150 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "per_async or (fast_path and True)" > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1; grep -E "^(FAILED|ERROR)|Fatal" $OUT/pytest.log | head -3 | cut -c1-200
