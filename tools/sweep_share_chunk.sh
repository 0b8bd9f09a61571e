# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
for c in 2560; do for sh in 80 85 90; do for i in 1 2; do
echo -n "chunk $c "; CORNETTO_SDUST_CHUNK=$c python bench.py --steps 30 --warmup 3 --no-profiles --no-e2e --no-cpu --no-reads --check-steps 0 --emulate-ranks= --sdust-share $sh 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('share', $sh, 'ms/step', d['ms_per_step'], d['ms_per_step_spread']['median'], d.get('stage_wall_ms'), {k: v['ms'] for k, v in d['kernels'].items() if v.get('ms', 0) > 0.25})"
done; done; done
