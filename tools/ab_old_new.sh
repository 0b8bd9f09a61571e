# same-box A/B: the working tree against a worktree of an older commit under _old/ (built there): bash tools/ab_old_new.sh "<gbases>" [reps]
B="--no-cpu --no-profiles --no-e2e --no-reads --no-second --emulate-ranks= --steps 40 --warmup 5"
for rep in $(seq ${2:-2}); do for gb in ${1:-3.16 0.395}; do for which in new old; do
  if [ $which = old ]; then BENCH=_old/bench.py; else BENCH=bench.py; fi
  timeout 200 python $BENCH $B --gbases $gb 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$which gbases=$gb', j['ms_per_step'], j['value'], j.get('stage_wall_ms'))
"
done; done; done
