# development aid: the bench step with an environment switch on / off, alternating on one box:   bash tools/ab_env.sh VAR ON OFF [share]
V=$1; ON=$2; OFF=$3; SH=${4:-76}
Q="--steps 30 --warmup 3 --no-cpu --no-e2e --no-reads --no-profiles --check-steps 0"
for val in $ON $OFF $ON $OFF; do
echo -n "$V=$val share $SH: "; env $V=$val python bench.py $Q --emulate-ranks 8 --sdust-share $SH 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['scaling_model']['8']
print('ms/step', d['ms_per_step'], d['ms_per_step_spread']['median'], d.get('stage_wall_ms'), '| 8:', m['step_ms'], m['efficiency'], m['stage_wall_ms_slowest'])"
done
