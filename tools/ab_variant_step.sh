# development aid: the bench step (and the stage walls) with the library as built against libcornetto_hip_variant.so (make -C cornetto_amd variant VARIANT_FLAGS=...), alternating on one box
V=$PWD/cornetto_amd/libcornetto_hip_variant.so
run() { echo -n "== $1: "; CORNETTO_LIB=$2 python bench.py --no-cpu --no-profiles --no-e2e --no-reads --emulate-ranks "" --steps 60 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print(d['ms_per_step'], 'share', d.get('sdust_share_percent'), 'sd_sift in step', k['sdust_kernel']['in_step']['ms'], 'cov_blocks', k['cov_blocks']['in_step']['ms'], 'tf_scan', k['tf_scan']['in_step']['ms'], d['stage_wall_ms'])"; }
for i in ${REPS:-1 2 3}; do
run "library" "" "$ARGS"
run "variant" $V "$ARGS"
done
