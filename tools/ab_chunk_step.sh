# development aid: the two-stream bench step against the sift chunk size (= LDS per wave = resident waves per CU)
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
Q="--steps 30 --warmup 3 --no-cpu --no-e2e --no-reads --no-profiles --check-steps 0 --emulate-ranks="
for c in ${@:-0 1408 0 1408}; do
echo -n "chunk $c: "; CORNETTO_SDUST_CHUNK=$c python bench.py $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'share', d.get('sdust_share_percent'), d.get('sdust_share_probe_ms'), d.get('stage_wall_ms'), {k: v['ms'] for k, v in d['kernels'].items() if v.get('ms', 0) > 0.25})"
done
