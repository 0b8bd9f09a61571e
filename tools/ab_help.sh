# development aid: the bench step with and without the helper launch of cornetto_accel_boost (CORNETTO_SDUST_HELP_BELOW: shares below it poll for the boost), full size and a 1/8 share
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
Q="--steps 30 --warmup 3 --no-cpu --no-e2e --no-reads --no-profiles --check-steps 0"
for sh in ${@:-72 76}; do for hb in 85 0 85 0; do
echo -n "share $sh help_below $hb: "; CORNETTO_SDUST_HELP_BELOW=$hb python bench.py $Q --emulate-ranks 8 --sdust-share $sh 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['scaling_model']['8']
print('ms/step', d['ms_per_step'], d['ms_per_step_spread']['median'], d.get('stage_wall_ms'), '| 8:', m['step_ms'], m['efficiency'], m['stage_wall_ms_slowest'])"
done; done
