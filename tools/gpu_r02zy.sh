#!/bin/bash
export TMPDIR=/tmp
for ks in 8 14 28; do echo "== DRA_FC4_KS=$ks"; DRA_FC4_KS=$ks timeout 200 python tests/diag_schedule.py async -1 normal 2>/dev/null | cut -c1-200; done
