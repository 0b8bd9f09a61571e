# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
for P in uniform humanlike; do
for cfg in "0 0" "1408 0" "1280 0" "0 6144" "0 5376"; do set -- $cfg
echo -n "$P chunk $1 blocks $2: "
if [ "$2" != "0" ]; then export CORNETTO_SIFT_BLOCKS=$2; else unset CORNETTO_SIFT_BLOCKS; fi
CORNETTO_SDUST_CHUNK=$1 python tools/perf_probe.py sdust --mbases 3160 --reps 4 --profile $P 2>&1 | grep -o "sdust_kernel., [0-9.]*\|digest [0-9a-f]*" | tail -4 | tr "\n" " "; echo
done; done
