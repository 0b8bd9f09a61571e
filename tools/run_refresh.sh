set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/tests.log
tail -3 gpurun_out/tests.log
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu > $R/gpurun_out/bench_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bgin -- python3 $R/tools/perf_bgin.py --mlines 60 --piece-mb 512 > $R/gpurun_out/bgin_prof.log 2>&1
tail -2 $R/gpurun_out/bgin_prof.log
