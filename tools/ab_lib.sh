# development aid: sd_sift alone, the library against libcornetto_hip_variant.so (make -C cornetto_amd variant VARIANT_FLAGS=...), alternating on one box
for P in ${PROFILES:-uniform humanlike satellite}; do for i in 1 2; do
for lib in "" $PWD/cornetto_amd/libcornetto_hip_variant.so; do
echo -n "$P ${lib:+variant}: "
CORNETTO_LIB=$lib python tools/perf_probe.py sdust --mbases ${MBASES:-3160} --reps 4 --profile $P 2>&1 | grep -o "sdust_kernel., [0-9.]*\|digest [0-9a-f]*" | tail -4 | tr "\n" " "; echo
done; done; done
