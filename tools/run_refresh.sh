#!/bin/bash
# end-of-round evidence: GPU tests, smoke, bench line, rocprofv3 kernel stats of the bench (two streams and serial, both workload
# profiles), SQ / traffic counters of the two sdust kernel families.   bash tools/run_refresh.sh [tag]
TAG=${1:-r06}
set -x
mkdir -p gpurun_out
R=$PWD
# the counters first: the bench line quotes them (roofline.issue / roofline.traffic read profiles/${TAG}_<profile>_*.json)
for P in uniform satellite humanlike; do
  bash tools/pmc_kernel.sh ${TAG}_$P sd_sift 3160 $P 1 | tail -12
  cp gpurun_out/${TAG}_${P}_sq_sd_sift.json gpurun_out/${TAG}_${P}_pmc_traffic_sd_sift.json profiles/ 2>/dev/null
done
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/${TAG}_tests.log
tail -3 gpurun_out/${TAG}_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
timeout 1200 python bench.py --steps 20 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 2 --serial --no-cpu --no-e2e --no-reads --no-profiles --emulate-ranks "" > gpurun_out/${TAG}_bench_serial.json 2>> gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
export CORNETTO_BENCH_WARM=0     # (no 4 kb warm-up launches in the traced runs: every sd_sift / cov_blocks / tf_scan row of the statistics is a full-size launch)
rm -rf $R/gpurun_out/prof_${TAG} $R/gpurun_out/prof_${TAG}_serial $R/gpurun_out/prof_${TAG}_sat $R/gpurun_out/prof_${TAG}_hum $R/gpurun_out/prof_${TAG}_share8
Q="--steps 5 --warmup 1 --no-cpu --no-e2e --no-reads --no-profiles --check-steps 0 --emulate-ranks="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py $Q > $R/gpurun_out/${TAG}_bench_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_serial -- python3 $R/bench.py $Q --serial > $R/gpurun_out/${TAG}_bench_prof_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_sat -- python3 $R/bench.py $Q --serial --profile satellite > $R/gpurun_out/${TAG}_bench_prof_sat.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_hum -- python3 $R/bench.py $Q --serial --profile humanlike > $R/gpurun_out/${TAG}_bench_prof_hum.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_share8 -- python3 $R/bench.py $Q --rank-share 8,6 > $R/gpurun_out/${TAG}_bench_prof_share8.log 2>&1
cd $R
python3 tools/summarize_profile.py stats gpurun_out/prof_${TAG} gpurun_out/${TAG}_kernel_stats.csv | head -12
python3 tools/summarize_profile.py stats gpurun_out/prof_${TAG}_serial gpurun_out/${TAG}_kernel_stats_serial.csv | head -12
python3 tools/summarize_profile.py stats gpurun_out/prof_${TAG}_sat gpurun_out/${TAG}_kernel_stats_satellite_serial.csv | head -8
python3 tools/summarize_profile.py stats gpurun_out/prof_${TAG}_hum gpurun_out/${TAG}_kernel_stats_humanlike_serial.csv | head -8
python3 tools/summarize_profile.py stats gpurun_out/prof_${TAG}_share8 gpurun_out/${TAG}_kernel_stats_share8.csv | head -30
rm -rf gpurun_out/prof_${TAG} gpurun_out/prof_${TAG}_serial gpurun_out/prof_${TAG}_sat gpurun_out/prof_${TAG}_hum gpurun_out/prof_${TAG}_share8
