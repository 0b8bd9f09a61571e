#!/bin/bash
# FETCH_SIZE of sd_sift calibrated in the kernel's OWN access pattern: the stages behind the staging switched off (CORNETTO_SIFT_ABL=7: the kernel
# still fetches every region — chunk + the 128 bases in front — exactly once, 1.0833 B/base for 1536-base chunks; 135: not even that), against the full kernel
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export CORNETTO_SDUST_SIFT=1
for abl in 0 7 135; do
  if [ "$abl" = "0" ]; then unset CORNETTO_SIFT_ABL; else export CORNETTO_SIFT_ABL=$abl; fi
  for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcf_x
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmcf_x --output-format csv -- python3 $R/tools/perf_probe.py sdust --mbases 3160 --features 1 --reps 2 --profile uniform > $R/gpurun_out/pmcf_x.log 2>&1
  python3 - $abl $R <<'PY'
import csv, glob, sys, collections
abl, R = sys.argv[1:3]
agg, n = collections.defaultdict(float), collections.defaultdict(set)
for f in glob.glob(R + "/gpurun_out/pmcf_x/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sd_sift" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for k, v in agg.items():
    per = v / max(1, len(n[k]))
    print("abl", abl, k, "KiB per launch", round(per, 1), "-> counted bytes per base", round(per * 1024 / 3160000088, 4))
PY
  done
done
rm -rf $R/gpurun_out/pmcf_x
