#!/bin/bash
# SQ counters and HBM traffic of one kernel of the sdust call (separate rocprofv3 --pmc passes, no other tracing):
#   bash tools/pmc_kernel.sh <tag> <kernel name substring> [mbases] [profile] [CORNETTO_SDUST_SIFT value]
#   -> gpurun_out/<tag>_sq_<substring>.json, gpurun_out/<tag>_pmc_traffic_<substring>.json      (copy them to profiles/)
TAG=${1:-r03}
KERN=${2:-sd_sift}
MB=${3:-3160}
PROFILE=${4:-uniform}
export CORNETTO_SDUST_SIFT=${5:-1}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PROBE="python3 $R/tools/perf_probe.py sdust --mbases $MB --features 1 --reps 2 --profile $PROFILE"
CORNETTO_LIB=$R/cornetto_amd/libcornetto_hip_dev.so CORNETTO_SDUST_STATS=1 $PROBE 2> $R/gpurun_out/${TAG}_stats.txt > $R/gpurun_out/${TAG}_probe.txt      # (the statistics build: a switch of the development library)
$PROBE 2> /dev/null > $R/gpurun_out/${TAG}_probe_prod.txt      # (the production build: the kernel the bench times)
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmck_${TAG}_s$i
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmck_${TAG}_s$i --output-format csv -- $PROBE > $R/gpurun_out/pmck_${TAG}_s$i.log 2>&1
done
cd $R
python3 - "$TAG" "$KERN" "$MB" "$PROFILE" "$CORNETTO_SDUST_SIFT" <<'PY'
import collections, csv, glob, json, sys, re, os
sys.path.insert(0, os.getcwd())
tag, kern, mb, profile, sift = sys.argv[1:6]
agg, n = collections.defaultdict(float), collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmck_%s_s*/**/*_counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
per = {k: v / max(1, len(n[k])) for k, v in agg.items()}
import bench
bases = sum(bench.contig_lengths(int(float(mb) * 1e6)))
stats = open("gpurun_out/%s_stats.txt" % tag).read()
probe = open("gpurun_out/%s_probe.txt" % tag).read()
m = re.findall(r"\('sdust_kernel', ([0-9.]+)\)", probe)
kernel_ms = float(m[-1]) if m else None
mp = re.findall(r"\('sdust_kernel', ([0-9.]+)\)", open("gpurun_out/%s_probe_prod.txt" % tag).read())
production_ms = float(mp[-1]) if mp else None
units = bases / 64.0
out = {"workload": "tools/perf_probe.py sdust --mbases %s --features 1 --profile %s, CORNETTO_SDUST_SIFT=%s (sdust alone on the chip)" % (mb, profile, sift), "kernel": kern,
       "stats_line": [l for l in stats.splitlines() if "sift:" in l or "wave-steps" in l][:2], "bases": bases, "kernel_ms": kernel_ms,
       "counting_build_ms": kernel_ms, "production_ms": production_ms,
       "kernel_ms_note": "kernel_ms = counting_build_ms: the build with the statistics counters (CORNETTO_SDUST_STATS=1), which the counter passes do NOT run; "
                         "production_ms: the kernel the counters were collected on and the bench times", "per_launch": per,
       "per_64_bases": {k: round(v / units, 2) for k, v in per.items() if k.startswith("SQ_INSTS") or k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS")}}
if kernel_ms and per.get("SQ_INSTS_VALU"):
    # 1024 SIMDs; a wave-64 vector instruction holds its SIMD's ALU for 4 cycles; the kernel's cycles from GRBM_GUI_ACTIVE (summed over 8 XCDs)
    cyc = per.get("GRBM_GUI_ACTIVE", 0) / 8.0
    kernel_ms = production_ms or kernel_ms          # (the counter passes run the production build)
    if cyc and cyc / (kernel_ms * 1e-3) / 1e9 < 1.0:
        # (the counter of this pass did not cover the kernel: several dispatches per pass; take the clock the sdust_w64 pass measured)
        out["clock_note"] = "GRBM_GUI_ACTIVE unusable for this kernel's passes: 2.2 GHz assumed (profiles/r03_sq_sdust_w64.json measured 2.14-2.21)"
        cyc = kernel_ms * 1e-3 * 2.2e9
    if cyc:
        out["valu_busy_of_kernel_time"] = round(4 * per["SQ_INSTS_VALU"] / (1024 * cyc), 4)
        out["clock_GHz_from_counters"] = round(cyc / (kernel_ms * 1e-3) / 1e9, 3)
json.dump(out, open("gpurun_out/%s_sq_%s.json" % (tag, kern), "w"), indent=1)
f, w = per.get("FETCH_SIZE"), per.get("WRITE_SIZE")
if f is not None and w is not None:
    # FETCH_SIZE / WRITE_SIZE count kilobytes; on gfx950 a coalesced stream is counted at half its bytes (profiles/r02_pmc_traffic.json:
    # tools/ubench/fetch_calib measured 2.000), the per-lane 32-byte requests of sdust_w64 at 1 / 1.68
    # (sd_sift: calibrated in the kernel's own access pattern, tools/pmc_fetch_calib_sift.sh -> profiles/r05_fetch_calib_sift.txt: 1.0833 known B/base read as 0.6236)
    scale = 1.737 if kern == "sd_sift" else 1.68
    tr = {kern: {"bases": bases, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "fetch_scale_used": scale, "fetch_bytes": f * 1024 * scale, "write_bytes": w * 1024,
                 "hbm_bytes": f * 1024 * scale + w * 1024, "bytes_per_base": round((f * 1024 * scale + w * 1024) / bases, 4)},
          "workload": out["workload"], "production_ms": production_ms, "calibration_note": "profiles/r02_pmc_traffic.json holds the calibration runs (tools/ubench/fetch_calib)"}
    json.dump(tr, open("gpurun_out/%s_pmc_traffic_%s.json" % (tag, kern), "w"), indent=1)
    print(json.dumps(tr[kern]))
print(json.dumps({k: out[k] for k in ("kernel_ms", "per_64_bases", "valu_busy_of_kernel_time", "clock_GHz_from_counters") if k in out}, indent=1))
PY
rm -rf gpurun_out/pmck_${TAG}_s*
