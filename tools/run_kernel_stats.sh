#!/bin/bash
# the rocprofv3 --kernel-trace --stats part of tools/run_refresh.sh alone: bash tools/run_kernel_stats.sh [tag]
TAG=${1:-r06}
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
export CORNETTO_BENCH_WARM=0     # (no 4 kb warm-up launches in the traced runs: every sd_sift / cov_blocks / tf_scan row of the statistics is a full-size launch)
rm -rf $R/gpurun_out/prof_${TAG} $R/gpurun_out/prof_${TAG}_serial $R/gpurun_out/prof_${TAG}_sat
Q="--steps 5 --warmup 1 --no-cpu --no-e2e --no-reads --no-profiles --check-steps 0 --emulate-ranks="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py $Q > $R/gpurun_out/${TAG}_bench_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_serial -- python3 $R/bench.py $Q --serial > $R/gpurun_out/${TAG}_bench_prof_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_sat -- python3 $R/bench.py $Q --serial --profile satellite > $R/gpurun_out/${TAG}_bench_prof_sat.log 2>&1
cd $R
python3 tools/summarize_profile.py stats gpurun_out/prof_${TAG} gpurun_out/${TAG}_kernel_stats.csv | head -14
python3 tools/summarize_profile.py stats gpurun_out/prof_${TAG}_serial gpurun_out/${TAG}_kernel_stats_serial.csv | head -8
python3 tools/summarize_profile.py stats gpurun_out/prof_${TAG}_sat gpurun_out/${TAG}_kernel_stats_satellite_serial.csv | head -5
