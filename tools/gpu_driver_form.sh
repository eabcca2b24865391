for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('run', d['value'], d['ms_per_step'], d['host'], d['long_run']['value'])"; done
