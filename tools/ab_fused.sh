# development aid: the bench step with the other thread as one call (cornetto_panel_step) against the three calls, full size and the 8-rank model, at fixed shares
Q="--steps 30 --warmup 3 --no-cpu --no-e2e --no-reads --no-profiles --check-steps 0"
for sh in ${@:-72 80}; do for f in 1 0 1 0; do
echo -n "share $sh fused $f: "; CORNETTO_BENCH_FUSED=$f python bench.py $Q --emulate-ranks 8 --sdust-share $sh 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['scaling_model']['8']
print('ms/step', d['ms_per_step'], d['ms_per_step_spread']['median'], d.get('stage_wall_ms'), '| 8:', m['step_ms'], m['efficiency'], m['stage_wall_ms_slowest'])"
done; done
