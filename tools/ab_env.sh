# bash tools/ab_env.sh "<ENV=VAL or ->" ... : the bench step at 3.16 Gbp with / without an environment switch, interleaved
B="--no-cpu --no-profiles --no-e2e --no-reads --no-second --emulate-ranks= --steps 40 --warmup 5 --gbases ${GB:-3.16}"
for rep in 1 2; do for e in "$@"; do
  if [ "$e" = "-" ]; then EV=""; else EV="$e"; fi
  env $EV timeout 200 python bench.py $B 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('[$e]', j['ms_per_step'], j['value'], j.get('stage_wall_ms'))
"
done; done
