set -x
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "lazy or boost" 2>&1 | tail -3
B="--no-cpu --no-profiles --no-e2e --no-reads --no-second --emulate-ranks= --steps 30 --warmup 5"
for cfg in "1 70" "0 70" "1 75" "1 80" "1 85" "1 70" "0 70" "1 80"; do set -- $cfg
  CORNETTO_BENCH_LAZY=$1 timeout 200 python bench.py $B --sdust-share $2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('LAZY=$1 share=$2', j['ms_per_step'], j['value'], j.get('stage_wall_ms'))
"
done
