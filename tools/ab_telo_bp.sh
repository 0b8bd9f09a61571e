# development aid: tf_scan as bit planes (CORNETTO_TF_BP=1, the default) against the shift-and automaton (0) on the development build of the library, alternating on one box:
# the telofind tests, the scan alone (tools/perf_probe.py telo) and the bench step
python -m pytest tests/test_gpu_parity.py -x -q -k "bit_planes" 2>&1 | grep -v "^$" | tail -40
D=$PWD/cornetto_amd/libcornetto_hip_dev.so
for i in 1 2; do for bp in 0 1; do
echo -n "== step, BP=$bp: "; CORNETTO_LIB=$D CORNETTO_TF_BP=$bp python bench.py --no-cpu --no-profiles --no-e2e --no-reads --emulate-ranks "" --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print(d['ms_per_step'], 'share', d.get('sdust_share_percent'), 'tf_scan in step', k['tf_scan']['in_step']['ms'], 'alone', k['tf_scan']['alone']['ms'], 'sd_sift in step', k['sdust_kernel']['in_step']['ms'], 'cov_blocks', k['cov_blocks']['in_step']['ms'], d['stage_wall_ms'])"
done; done
