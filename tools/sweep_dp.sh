# development aid: sd_sift alone against the dp-tile threshold and the L2-skip threshold (scheduling choices: results do not change)
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
for P in ${PROFILES:-humanlike satellite}; do
for cfg in "24 48" "16 48" "20 48" "32 48" "24 32" "24 40" "24 56" "24 65"; do set -- $cfg
echo -n "$P dp>=$1 l2skip>=$2: "
CORNETTO_SIFT_DP=$1 CORNETTO_SIFT_L2SKIP=$2 python tools/perf_probe.py sdust --mbases 3160 --reps 3 --profile $P 2>&1 | grep -o "sdust_kernel., [0-9.]*" | tail -2 | tr "\n" " "; echo
done; done
