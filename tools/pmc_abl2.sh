#!/bin/bash
# per-stage instruction and LDS-cycle counts of sd_sift (CORNETTO_SIFT_ABL: 4 no tiles, 2 no L1 / L2, 1 no resolve): bash tools/pmc_abl2.sh [profile] [abl values]
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
R=$PWD
P=${1:-uniform}; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export CORNETTO_SDUST_SIFT=1
for abl in ${@:-0 1 3 7}; do
  if [ "$abl" = "0" ]; then unset CORNETTO_SIFT_ABL; elif [ "$abl" = "g" ]; then unset CORNETTO_SIFT_ABL; export CORNETTO_SIFT_GENERIC=1; else export CORNETTO_SIFT_ABL=$abl; fi   # (0: the production build; g: the build for run-time buffer sizes; other values run that build with stages off)
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  rm -rf $R/gpurun_out/pmcabl_x
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmcabl_x --output-format csv -- python3 $R/tools/perf_probe.py sdust --mbases 3160 --features 1 --reps 2 --profile $P > $R/gpurun_out/pmcabl_x.log 2>&1
  python3 - $abl $R <<'PY'
import csv, glob, sys, collections
abl, R = sys.argv[1:3]
agg, n = collections.defaultdict(float), collections.defaultdict(set)
for f in glob.glob(R + "/gpurun_out/pmcabl_x/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sd_sift" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print("abl", abl, {k: round(v / max(1, len(n[k])) / (3160000088 / 64), 2) for k, v in sorted(agg.items())}, "per 64 bases")
PY
  done
done
rm -rf $R/gpurun_out/pmcabl_x
