# bash tools/ab_model.sh "<ENV=VAL or ->" ...: the strong-scaling model (8 shares of the 3.16 Gbp assembly, one after the other) with / without a switch
B="--no-cpu --no-profiles --no-e2e --no-reads --no-second --emulate-ranks 8 --steps 20 --warmup 5"
cat > /tmp/ab_model_fmt.py <<'PY'
import sys, json
tag = sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); m = j['scaling_model']['8']
        print('[%s]' % tag, j['ms_per_step'], m['step_ms'], m['efficiency'], m['per_rank_ms'], m.get('contigs_per_rank'))
        print('   slowest', m.get('stage_wall_ms_slowest')); print('   fastest', m.get('stage_wall_ms_fastest'))
PY
for rep in 1 2; do for e in "$@"; do
  if [ "$e" = "-" ]; then EV=""; else EV="$e"; fi
  env $EV timeout 300 python bench.py $B 2>/dev/null | python /tmp/ab_model_fmt.py "$e"
done; done
