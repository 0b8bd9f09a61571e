#!/usr/bin/env python3
"""condense the rocprofv3 --pmc passes of tools/pmc_sdust.sh (gpurun_out/pmc_<tag>_*) into gpurun_out/<tag>_sq_sdust.json and
gpurun_out/<tag>_pmc_traffic.json:   python3 tools/pmc_summarize.py <tag> <mbases> <profile>"""
import collections, csv, glob, json, re, sys
tag, mb, profile = sys.argv[1], float(sys.argv[2]), sys.argv[3]
def counters(pattern, want):
    agg, n = collections.defaultdict(float), collections.defaultdict(set)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            for w in want:
                if w in r["Kernel_Name"]:
                    agg[(w, r["Counter_Name"])] += float(r["Counter_Value"])
                    n[(w, r["Counter_Name"])].add((f, r["Dispatch_Id"]))
    return {k: v / max(1, len(n[k])) for k, v in agg.items()}          # per launch
sq = counters("gpurun_out/pmc_%s_s*/**/*_counter_collection.csv" % tag, ["sdust_w64", "sdust_dense"])
stats = open("gpurun_out/%s_stats.txt" % tag).read()
m = re.search(r"wave-steps (\d+) find_perfect calls (\d+) \((\d+) with candidates\) plain groups (\d+)", stats)
steps = int(m.group(1)) if m else 0
out = {"workload": "tools/perf_probe.py sdust --mbases %g --features 1 --profile %s (the bench assembly, sdust alone on the chip)" % (mb, profile),
       "wave_steps": steps, "find_perfect_calls": int(m.group(2)) if m else None, "find_perfect_with_candidates": int(m.group(3)) if m else None,
       "plain_groups_of_4_steps": int(m.group(4)) if m else None, "per_launch": {}, "per_wave_step": {}}
for (k, c), v in sorted(sq.items()):
    out["per_launch"].setdefault(k, {})[c] = v
    if k == "sdust_w64" and steps and (c.startswith("SQ_INSTS") or c == "SQ_ACTIVE_INST_MISC"):
        out["per_wave_step"][c] = round(v / steps, 2)
w = out["per_launch"].get("sdust_w64", {})
if w.get("SQ_BUSY_CYCLES") and w.get("SQ_ACTIVE_INST_VALU"):
    out["valu_active_quadcycles_x4_over_busy_cycles_per_simd"] = round(4 * w["SQ_ACTIVE_INST_VALU"] / (w["SQ_BUSY_CYCLES"] * 4 / 4), 4)
json.dump(out, open("gpurun_out/%s_sq_sdust.json" % tag, "w"), indent=1)
print(json.dumps(out["per_wave_step"]), steps)
# ---- traffic, with the calibration of FETCH_SIZE on this access pattern
cal = counters("gpurun_out/pmc_%s_calib/**/*_counter_collection.csv" % tag, ["calib_coalesced", "calib_lane<2>", "calib_lane<4>"])
log = open("gpurun_out/pmc_%s_calib.log" % tag).read()
b_co = int(re.search(r"calib_coalesced (\d+)", log).group(1)) if "calib_coalesced" in log else 0
b_ln = int(re.search(r"each calib_lane (\d+)", log).group(1)) if "each calib_lane" in log else 0
calib = {}
for k, b in (("calib_coalesced", b_co), ("calib_lane<2>", b_ln), ("calib_lane<4>", b_ln)):
    v = cal.get((k, "FETCH_SIZE"))
    if v and b:
        calib[k] = {"bytes_read": b, "FETCH_SIZE_KiB": v, "bytes_per_counted_byte": round(b / (v * 1024), 4)}
fe = counters("gpurun_out/pmc_%s_fetch/**/*_counter_collection.csv" % tag, ["sdust_w64", "sdust_dense"])
wr = counters("gpurun_out/pmc_%s_write/**/*_counter_collection.csv" % tag, ["sdust_w64", "sdust_dense"])
bases = None
for l in open("gpurun_out/pmc_%s_fetch.log" % tag):
    pass
import os, sys
sys.path.insert(0, os.getcwd())
import bench
bases = sum(bench.contig_lengths(int(mb * 1e6)))
tr = {"workload": out["workload"], "bases": bases, "calibration": calib,
      "calibration_note": "tools/ubench/fetch_calib: 4 GiB read exactly once per kernel; bytes_per_counted_byte = true bytes / (FETCH_SIZE x 1024). "
                          "calib_lane<2> is the access pattern of sdust_w64 (every lane its own region, 32 bytes per request group)"}
f = fe.get(("sdust_w64", "FETCH_SIZE"))
wv = wr.get(("sdust_w64", "WRITE_SIZE"))
if f is not None and wv is not None:
    scale = calib.get("calib_lane<2>", {}).get("bytes_per_counted_byte", 2.0)
    tr["sdust_w64"] = {"bases": bases, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": wv, "fetch_scale_used": scale,
                       "fetch_bytes": f * 1024 * scale, "write_bytes": wv * 1024, "hbm_bytes": f * 1024 * scale + wv * 1024,
                       "bytes_per_base": round((f * 1024 * scale + wv * 1024) / bases, 4)}
json.dump(tr, open("gpurun_out/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(json.dumps(tr.get("calibration")), json.dumps(tr.get("sdust_w64")))
