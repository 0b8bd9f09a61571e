# development aid: sdust kernel time against chunk size and the number of queue passes
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
for cfg in "0 8" "896 8" "896 16" "896 32" "448 32" "448 64" "1280 16" "2560 8" "0 8"; do set -- $cfg
echo -n "chunk $1 passes $2: "
CORNETTO_SDUST_CHUNK=$1 CORNETTO_SDUST_PASSES=$2 python tools/perf_probe.py sdust --mbases 3160 --reps 4 2>&1 | grep -o "sdust_kernel., [0-9.]*\|digest.*" | tail -6 | tr "\n" " "; echo
done
