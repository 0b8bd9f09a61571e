cd /root/repo
R=$PWD
cd /tmp && export TMPDIR=/tmp
for abl in 7 3 1 0; do
  export CORNETTO_SIFT_ABL=$abl
  PROBE="python3 $R/tools/perf_probe.py sdust --mbases 3160 --features 1 --reps 1 --profile uniform"
  rm -rf /tmp/pm_$abl; 
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH -d /tmp/pm_$abl --output-format csv -- $PROBE > /dev/null 2>&1
  python3 - $abl <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(float)
for f in glob.glob("/tmp/pm_%s/**/*_counter_collection.csv"%sys.argv[1],recursive=True):
    for r in csv.DictReader(open(f)):
        if "sd_sift" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
print("abl",sys.argv[1],{k: round(v/52.9e6,1) for k,v in sorted(agg.items())})
PY
done
