# development aid: sd_sift alone on the chip against the number of resident waves per CU (CORNETTO_SIFT_BLOCKS = waves per CU x 256)
# and against the chunk size (which sets the LDS of a wave: 1280-byte granules).   bash tools/sweep_waves.sh [profile]
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
P=${1:-uniform}
for w in 8 12 16 18 20 21; do
echo -n "waves/CU $w: "
CORNETTO_SIFT_BLOCKS=$((w*256)) python tools/perf_probe.py sdust --mbases 3160 --reps 4 --profile $P 2>&1 | grep -o "sdust_kernel., [0-9.]*" | tail -3 | tr "\n" " "; echo
done
for c in 1088 1280 1536 2048; do
echo -n "chunk $c: "
CORNETTO_SDUST_CHUNK=$c python tools/perf_probe.py sdust --mbases 3160 --reps 4 --profile $P 2>&1 | grep -o "sdust_kernel., [0-9.]*" | tail -3 | tr "\n" " "; echo
done
