#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/<dir>/.../*_kernel_stats.csv and *_counter_collection.csv) into the
small, tracked summaries under profiles/.

    python tools/summarize_profile.py stats  gpurun_out/prof_r01  profiles/r01_kernel_stats.csv
    python tools/summarize_profile.py pmc    gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json
"""
import collections
import csv
import glob
import json
import sys

OURS = ("sd_sift", "sd_sample_count", "sdust_w64", "sdust_dense", "sdust_kernel", "sdust_gather", "tf_scan", "tf_gather", "tf_pair", "tf_ctgoff", "tf_greedy", "tw_scan", "tw_fill",
        "cov_blocks", "cov_windows", "cov_order", "cov_total64", "scan_local", "scan_partials", "scan_add",
        "tk_count", "tk_scatter", "bg_records", "bg_layout", "sd_wordcount",
        "fq_nl_count", "fq_nl_scatter", "fq_records", "fq_pack", "fa_lines", "fa_heads", "fa_records", "fa_linedst", "fa_copy", "sd_prep", "sd_order", "st_local", "st_emit",
        "scan_lookback", "st_fused", "cov_ctg_first", "sd_make_order")


def short(name):
    for k in OURS:
        if k in name:
            return k
    return None


def newest(files):
    """gpurun merges every call's output into the same directory: keep the files of the most recent run only"""
    import os
    if not files:
        return files
    t = max(os.path.getmtime(f) for f in files)
    return [f for f in files if t - os.path.getmtime(f) < 60]


def stats(src, dst):
    rows = []
    for f in newest(glob.glob(src + "/**/*_kernel_stats.csv", recursive=True)):
        rows += list(csv.DictReader(open(f)))
    with open(dst, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct_of_gpu_time"])
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            k = short(r["Name"])
            if not k:
                continue
            w.writerow([k, r["Calls"], "%.4f" % (float(r["TotalDurationNs"]) / 1e6), "%.4f" % (float(r["AverageNs"]) / 1e6),
                        "%.4f" % (float(r["MinNs"]) / 1e6), "%.4f" % (float(r["MaxNs"]) / 1e6), r["Percentage"]])
    print(open(dst).read())


def pmc(fetch_dir, write_dir, dst):
    out = {}
    for d, counter in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        agg = collections.defaultdict(float)
        disp = collections.defaultdict(set)
        for f in newest(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k and r["Counter_Name"] == counter:
                    agg[k] += float(r["Counter_Value"])
                    disp[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            out.setdefault(k, {})[counter + "_KB_per_launch"] = v / max(1, len(disp[k]))
    res = {}
    for k, v in out.items():
        f = v.get("FETCH_SIZE_KB_per_launch", 0.0) * 1024
        wr = v.get("WRITE_SIZE_KB_per_launch", 0.0) * 1024
        # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced stream
        # (128-B requests tallied at 64 B): doubled here.  WRITE_SIZE is exact for 16-B-per-lane stores.
        res[k] = {"fetch_bytes_raw": f, "fetch_bytes_corrected_x2": 2 * f, "write_bytes": wr, "hbm_bytes_per_launch": 2 * f + wr}
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
