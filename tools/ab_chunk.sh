# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
B="--serial --no-cpu --no-profiles --no-e2e --no-reads --no-second --emulate-ranks= --steps 20 --warmup 3"
for ch in 1152 1536 1024 1152 1280; do
  CORNETTO_SDUST_CHUNK=$ch timeout 200 python bench.py $B 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('chunk=$ch', j['ms_per_step'], j['kernels']['sdust_kernel'])
"
done
