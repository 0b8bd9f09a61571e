# development aid: sd_sift alone (uniform / humanlike), default chunk and 2048, with the kernel's own statistics once
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
for P in uniform humanlike; do
for c in 0 2048; do
echo -n "$P chunk $c: "
CORNETTO_SDUST_CHUNK=$c python tools/perf_probe.py sdust --mbases 3160 --reps 4 --profile $P 2>&1 | grep -o "sdust_kernel., [0-9.]*\|digest [0-9a-f]*" | tail -4 | tr "\n" " "; echo
done
CORNETTO_SDUST_STATS=1 python tools/perf_probe.py sdust --mbases 3160 --reps 1 --profile $P 2>&1 | grep "sdust stats. sift" | head -1
done
