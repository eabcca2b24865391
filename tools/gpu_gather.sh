export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "ring or replay or gather" 2>&1 | tail -3
python tools/bench_kernels.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k in ('gather_k1', 'gather_k64', 'gather_k1024'):
    print(k, round(d[k]['seconds'] * 1e6, 1), 'us', round(d[k]['read_plus_write_GBps']), 'GB/s', round(d[k]['frac_hbm_total_8TBps'], 3))"
