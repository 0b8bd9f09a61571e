# development aid: sd_sift alone against the chunk size (per-chunk fixed costs against LDS per wave), uniform and humanlike
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
for P in ${PROFILES:-uniform humanlike}; do
for c in ${@:-0 2048 2560 3072 3584}; do
echo -n "$P chunk $c: "
CORNETTO_SDUST_CHUNK=$c python tools/perf_probe.py sdust --mbases 3160 --reps 4 --profile $P 2>&1 | grep -o "sdust_kernel., [0-9.]*" | tail -3 | tr "\n" " "; echo
done; done
