# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PROBE="python3 $R/tools/perf_probe.py sdust --mbases 3160 --features 1 --reps 1"
CORNETTO_SDUST_STATS=1 $PROBE 2>&1 | grep -a "wave-steps"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/mini_s$i
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/mini_s$i --output-format csv -- $PROBE > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(float)
for f in glob.glob("gpurun_out/mini_s*/**/*_counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "sdust_w64" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
print({k: round(v / 1e6, 1) for k, v in agg.items()})
PY
rm -rf gpurun_out/mini_s*
