#!/bin/bash
# development aid: the ceiling of folding the telomere scan into the sdust waves — the bench step WITHOUT telo_scan against the step as it is, alternating on one box
A="--steps 20 --warmup 5 --no-cpu --no-profiles --no-e2e --no-reads --emulate-ranks= --check-steps 0"
for i in 1 2 3; do
  for nt in 0 1; do
    echo -n "no_telo=$nt: "
    CORNETTO_BENCH_NO_TELO=$nt python bench.py $A 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], 'ms; share', d['sdust_share_percent'], '; sdust_kernel in step', d['kernels']['sdust_kernel']['ms'], '; stages', d['stage_wall_ms'])"
  done
done
