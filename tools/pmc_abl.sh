#!/bin/bash
# instruction counts of sd_sift with stages switched off (CORNETTO_SIFT_ABL: 4 = no tiles, 2 = no L1 / L2, 1 = no resolve): where the instructions go
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export CORNETTO_SDUST_SIFT=1
for abl in ${@:-0 4 1 3}; do
  export CORNETTO_SIFT_ABL=$abl
  rm -rf $R/gpurun_out/pmcabl_$abl
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH -d $R/gpurun_out/pmcabl_$abl --output-format csv -- python3 $R/tools/perf_probe.py sdust --mbases 3160 --features ${FEATURES:-1} --reps 2 --profile uniform > $R/gpurun_out/pmcabl_$abl.log 2>&1
  python3 - $abl $R <<'PY'
import csv, glob, sys, collections
abl, R = sys.argv[1:3]
agg, n = collections.defaultdict(float), collections.defaultdict(set)
for f in glob.glob(R + "/gpurun_out/pmcabl_%s/**/*_counter_collection.csv" % abl, recursive=True):
    for r in csv.DictReader(open(f)):
        if "sd_sift" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print("abl", abl, {k: round(v / max(1, len(n[k])) / (3160000088 / 64), 2) for k, v in sorted(agg.items())}, "per 64 bases")
PY
done
