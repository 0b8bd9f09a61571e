#!/bin/bash
# SQ counters of one kernel of the sdust call (separate rocprofv3 --pmc passes, no other tracing):
#   bash tools/pmc_kernel.sh <tag> <kernel name substring> [mbases] [profile]     -> gpurun_out/<tag>_sq_<substring>.json
TAG=${1:-r03}
KERN=${2:-sd_sift}
MB=${3:-3160}
PROFILE=${4:-uniform}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PROBE="python3 $R/tools/perf_probe.py sdust --mbases $MB --features 1 --reps 1 --profile $PROFILE"
CORNETTO_SDUST_STATS=1 $PROBE 2> $R/gpurun_out/${TAG}_stats.txt > /dev/null
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES" "GRBM_GUI_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmck_${TAG}_s$i
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmck_${TAG}_s$i --output-format csv -- $PROBE > $R/gpurun_out/pmck_${TAG}_s$i.log 2>&1
done
cd $R
python3 - "$TAG" "$KERN" "$MB" "$PROFILE" <<'PY'
import collections, csv, glob, json, sys, re
tag, kern, mb, profile = sys.argv[1:5]
agg, n = collections.defaultdict(float), collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmck_%s_s*/**/*_counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
per = {k: v / max(1, len(n[k])) for k, v in agg.items()}
stats = open("gpurun_out/%s_stats.txt" % tag).read()
m = re.search(r"sift: tiles (\d+)", stats)
tiles = int(m.group(1)) if m else 0
out = {"workload": "tools/perf_probe.py sdust --mbases %s --features 1 --profile %s (sdust alone on the chip)" % (mb, profile), "kernel": kern,
       "stats_line": [l for l in stats.splitlines() if "sift:" in l][:1], "tiles": tiles, "per_launch": per,
       "per_tile": {k: round(v / tiles, 2) for k, v in per.items() if tiles and (k.startswith("SQ_INSTS") or k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS"))}}
if per.get("SQ_BUSY_CYCLES") and per.get("SQ_ACTIVE_INST_VALU"):
    out["valu_busy"] = round(4 * per["SQ_ACTIVE_INST_VALU"] / (per["SQ_BUSY_CYCLES"] * 4), 4)
json.dump(out, open("gpurun_out/%s_sq_%s.json" % (tag, kern), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/pmck_${TAG}_s*
