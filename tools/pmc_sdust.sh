#!/bin/bash
# development aid: SQ counters of the sdust kernel (separate rocprofv3 --pmc passes), summarised per wave-step
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
MB=${1:-500}
CORNETTO_SDUST_STATS=1 python3 $R/tools/perf_probe.py sdust --mbases $MB --features 1 --reps 1 2>&1 | grep -i "stats\|sdust" | tail -3
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmcs$i --output-format csv -- python3 $R/tools/perf_probe.py sdust --mbases $MB --features 1 --reps 1 > $R/gpurun_out/pmcs$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmcs*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sdust_w64" in r["Kernel_Name"] or "sdust_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print("%-26s %.4g" % (k, v))
PY
