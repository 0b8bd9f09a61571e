# bash tools/ab_lazy.sh "<lazy:share pairs>": the bench step with / without lazy result copies at a few sdust shares
B="--no-cpu --no-profiles --no-e2e --no-reads --no-second --emulate-ranks= --steps 30 --warmup 5"
for cfg in ${1:-1:70 0:70 1:75 1:65}; do lz=${cfg%%:*}; sh=${cfg##*:}
  CORNETTO_BENCH_LAZY=$lz timeout 200 python bench.py $B --sdust-share $sh 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('LAZY=$lz share=$sh', j['ms_per_step'], j['value'], j.get('stage_wall_ms'))
"
done
