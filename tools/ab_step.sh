# bash tools/ab_step.sh "<gbases values>" [reps]: the two-stream bench step at a few sizes
B="--no-cpu --no-profiles --no-e2e --no-reads --no-second --emulate-ranks= --steps 40 --warmup 5"
for rep in $(seq ${2:-2}); do for gb in ${1:-3.16 0.395}; do
  timeout 200 python bench.py $B --gbases $gb 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('gbases=$gb', j['ms_per_step'], j['value'], j.get('stage_wall_ms'))
"
done; done
