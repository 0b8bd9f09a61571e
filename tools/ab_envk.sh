# bash tools/ab_envk.sh "<ENV=VAL or ->" ...: sdust alone (bench --serial, 3.16 Gbp) with / without an environment switch, interleaved
B="--serial --no-cpu --no-profiles --no-e2e --no-reads --no-second --emulate-ranks= --steps 20 --warmup 3"
for rep in 1 2 3; do for e in "$@"; do
  if [ "$e" = "-" ]; then EV=""; else EV="$e"; fi
  env $EV timeout 200 python bench.py $B 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('[$e]', j['ms_per_step'], j['kernels']['sdust_kernel'])
"
done; done
