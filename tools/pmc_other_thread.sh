#!/bin/bash
# what the OTHER thread's kernels of a bench step issue (vector / scalar / LDS instructions, LDS-array cycles) per 64 bases — the pipes
# sd_sift is bound on.  Serial step, separate --pmc passes.   bash tools/pmc_other_thread.sh <tag>   -> gpurun_out/<tag>_sq_other_thread.json
TAG=${1:-r05}
R=$PWD
cd /tmp && export TMPDIR=/tmp
Q="all --mbases 3160 --features 1 --reps 2 --simple-cov 1"        # (rocprofv3 --pmc crashes inside torch.poisson: uniform random depth)
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmco_${TAG}_s$i
  timeout 400 rocprofv3 --pmc $set -d $R/gpurun_out/pmco_${TAG}_s$i --output-format csv -- python3 $R/tools/perf_probe.py $Q > $R/gpurun_out/pmco_${TAG}_s$i.log 2>&1
done
cd $R
python3 - "$TAG" <<'PY'
import collections, csv, glob, json, sys, os
sys.path.insert(0, os.getcwd())
tag = sys.argv[1]
import bench
bases = sum(bench.contig_lengths(int(3160e6)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pmco_%s_s*/**/*_counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].split()[-1]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
out = {"workload": "tools/perf_probe.py all --mbases 3160 --simple-cov 1 (uniform 3.16 Gbp, stages one after the other), per launch / per 64 bases", "bases": bases}
for k in sorted(agg):
    per = {c: v / max(1, n[k][c]) for c, v in agg[k].items()}
    if per.get("SQ_INSTS_VALU", 0) < 1e6: continue
    out[k] = {"launches_seen": max(n[k].values()), "per_64_bases": {c: round(v / (bases / 64.0), 3) for c, v in per.items() if c.startswith("SQ_INSTS") or c.startswith("SQ_LDS") or c == "SQ_ACTIVE_INST_LDS"}}
json.dump(out, open("gpurun_out/%s_sq_other_thread.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/pmco_${TAG}_s*
