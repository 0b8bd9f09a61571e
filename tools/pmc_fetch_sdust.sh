R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmcf --output-format csv -- python3 $R/tools/perf_probe.py sdust --mbases 3160 --features 0 --reps 1 > $R/gpurun_out/pmcf.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
tot = 0.0
for f in glob.glob("gpurun_out/pmcf/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sdust_w64" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"])
print("sdust_w64 FETCH_SIZE raw B/base: %.3f" % (tot * 1024 / 3160108082))
PY
timeout 300 python3 tools/perf_probe.py sdust --mbases 3160 --features 1 --reps 4 2>&1 | tail -2 | cut -c1-60
