# bash tools/ab_abl.sh "<abl values>" "<gbases values>": sdust alone (bench --serial) with CORNETTO_SIFT_ABL bits set
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
B="--serial --no-cpu --no-profiles --no-e2e --no-reads --no-second --emulate-ranks= --steps 20 --warmup 3"
for abl in ${1:-0}; do for gb in ${2:-3.16}; do
  CORNETTO_SIFT_ABL=$abl timeout 200 python bench.py $B --gbases $gb 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('abl=$abl gbases=$gb', j['ms_per_step'], j['kernels']['sdust_kernel'])
"
done; done
