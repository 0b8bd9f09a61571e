Q="--steps 30 --warmup 3 --no-cpu --no-e2e --no-reads --no-profiles --emulate-ranks 8"
for r in 1 2; do for v in 0 1; do
echo -n "fused $v: "
CORNETTO_BENCH_FUSED=$v python bench.py $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['scaling_model']['8']
print('ms/step', d['ms_per_step'], d['ms_per_step_spread']['median'], d['sdust_share_percent'], {a:round(b,2) for a,b in d.get('stage_wall_ms').items()}, d['determinism']['identical'], '| 8:', m['step_ms'], m['efficiency'], m['stage_wall_ms_slowest'])"
done; done
