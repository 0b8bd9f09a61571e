#!/bin/bash
# end-of-round evidence (round 2): GPU tests, smoke, bench line, rocprofv3 kernel stats of the bench (two streams and serial),
# SQ / traffic counters of the sdust kernel on both workload profiles.   bash tools/run_refresh.sh [tag]
TAG=${1:-r02}
set -x
mkdir -p gpurun_out
R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/${TAG}_tests.log
tail -3 gpurun_out/${TAG}_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py --steps 20 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 2 --serial --no-cpu --no-e2e --no-profiles > gpurun_out/${TAG}_bench_serial.json 2>> gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG} $R/gpurun_out/prof_${TAG}_serial $R/gpurun_out/prof_${TAG}_sat
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-e2e --no-profiles --check-steps 0 > $R/gpurun_out/${TAG}_bench_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_serial -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-e2e --no-profiles --check-steps 0 --serial > $R/gpurun_out/${TAG}_bench_prof_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_sat -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-e2e --no-profiles --check-steps 0 --serial --profile satellite > $R/gpurun_out/${TAG}_bench_prof_sat.log 2>&1
cd $R
python3 tools/summarize_profile.py stats gpurun_out/prof_${TAG} gpurun_out/${TAG}_kernel_stats.csv | head -12
python3 tools/summarize_profile.py stats gpurun_out/prof_${TAG}_serial gpurun_out/${TAG}_kernel_stats_serial.csv | head -12
python3 tools/summarize_profile.py stats gpurun_out/prof_${TAG}_sat gpurun_out/${TAG}_kernel_stats_satellite_serial.csv | head -8
bash tools/pmc_sdust.sh ${TAG} 3160 uniform | tail -3
bash tools/pmc_sdust.sh ${TAG}sat 3160 satellite | tail -3
