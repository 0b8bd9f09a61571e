B="--no-cpu --no-profiles --no-e2e --no-reads --no-second --emulate-ranks= --steps 40 --warmup 5 --gbases 0.395"
for cfg in "1 70" "0 70" "1 100" "1 50"; do set -- $cfg
  CORNETTO_BENCH_LAZY=$1 timeout 200 python bench.py $B --sdust-share $2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('LAZY=$1 share=$2', j['ms_per_step'], j['value'], j.get('stage_wall_ms')); print({k:v.get('ms') for k,v in j['kernels'].items() if v.get('ms')})
"
done
timeout 200 python bench.py $B --serial --timing 2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('serial', j['ms_per_step'], j.get('stage_wall_ms')); print({k:v.get('ms') for k,v in j['kernels'].items() if v.get('ms')})
"
