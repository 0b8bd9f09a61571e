#!/bin/bash
# end-of-round evidence: GPU tests, bench line, rocprofv3 kernel stats of the bench, PMC traffic passes
set -x
mkdir -p gpurun_out
R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/tests.log
tail -3 gpurun_out/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.json | cut -c1-400
timeout 600 python bench.py --steps 10 --warmup 2 --serial --no-cpu > gpurun_out/bench_serial.json 2>> gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu > $R/gpurun_out/bench_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01_serial -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --serial > $R/gpurun_out/bench_prof_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bgin -- python3 $R/tools/perf_bgin.py --mlines 60 --piece-mb 512 > $R/gpurun_out/bgin_prof.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/perf_probe.py all --mbases 3160 --features 0 --reps 1 --simple-cov 1 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/perf_probe.py all --mbases 3160 --features 0 --reps 1 --simple-cov 1 > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fastq -- python3 $R/tools/perf_fastq.py --mbases 1000 --read-len 10000 --reps 3 > $R/gpurun_out/fastq_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fasta -- python3 $R/tools/perf_fasta.py --width 80 --reps 3 > $R/gpurun_out/fasta_prof.log 2>&1
tail -2 $R/gpurun_out/bgin_prof.log
