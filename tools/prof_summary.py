#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one small text summary that is
committed under profiles/.  usage: prof_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.replace("knox::", "")
    return name[:90]


print("# rocprofv3 summary of `python3 bench.py --no-cpu-baseline --no-placement-ab` = the default run, 20 steps + 3 warmup, headline + configs[2] / configs[3] (%s)" % os.path.basename(root))
print("# scan_kernel<8, 1, 1u, false, false, 0u> = the scans: 12 of bench.py's one setup check + 3 warmup + 20 timed + 30 of the 0 % / 100 % selectivity runs;")
print("# <..., 1u> (FLV 1) = the arena's probes (chunk labels, write-buffer verification: alternating read-only / scan launches over 1 GiB samples);")
print("# scan_kernel<8, 1, 0u, ...> = the read-only form (setup check, ceiling after the timed region) AND the uint64 EQ + count scans of configs[3];")
print("# scan_kernel<8, 2, 14u, false, true, 2u> = configs[2] (2 x float64, AND, min / max / mean); bytes_eq_kernel = configs[3] []byte leg")
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("\n## kernel stats (--kernel-trace --stats)")
    print("%-92s %7s %12s %12s %8s" % ("kernel", "calls", "avg_ns", "total_ns", "pct"))
    for r in csv.DictReader(open(f)):
        print("%-92s %7s %12s %12s %8s" % (short(r["Name"]), r["Calls"], r["AverageNs"], r["TotalDurationNs"], r["Percentage"]))
# Kernels that run both over the whole column and over the few packs a bloom probe leaves (configs[3]): the kernel-stats
# average mixes them; the full-size launches are the dispatches whose duration is at least half of the longest one.
SPLIT = {"bytes_eq_kernel": "configs[3] []byte leg, full scans", "scan_kernel<8, 1, 0u, false, false, 0u>": "uint64 EQ / read-only form, full scans"}
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    dur = defaultdict(list)
    for r in csv.DictReader(open(f)):
        nm = short(r["Kernel_Name"])
        for k in SPLIT:
            if k in nm:
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if dur:
        print("\n## full-size launches only (duration >= 1/2 of the longest dispatch of that kernel)")
    for k, v in dur.items():
        big = [x for x in v if x >= max(v) / 2]
        print("%-60s %4d of %4d launches, avg_ns %12.1f  (%s)" % (k, len(big), len(v), sum(big) / len(big), SPLIT[k]))
tj = os.path.join(root, "trace_bench.json")
if os.path.exists(tj):
    print("\n## bench line under the profiler\n" + open(tj).read().strip())
# kernels whose HBM traffic is reported (label -> substring of the demangled name)
TRAFFIC_KERNELS = {
    "headline": "scan_kernel<8, 1, 1u, false, false, 0u>",        # int64 Between + sum (BASELINE configs[1])
    "cfg3": "scan_kernel<8, 2, 14u, false, true, 2u>",             # 2 x float64, AND, min / max / mean (configs[2])
    "cfg4_bytes": "bytes_eq_kernel",                                # []byte EQ over per-pack StringPools (configs[3])
    "cfg4_u64": "scan_kernel<8, 1, 0u, false, false, 0u>",         # uint64 EQ + count (also the read-only form: full-size dispatches)
    "u8_filter_sum": "scan_kernel<1, 1, 1u, false, false, 0u>",     # tools/bench_narrow.py: 1- / 2-byte filter + sum
    "i16_filter_sum": "scan_kernel<2, 1, 1u, false, false, 0u>",
    "bloom_count_u64": "bloom_count_kernel<8, HashNumeric>",
    "bloom_bin_u64": "bloom_bin_kernel<8, HashNumeric>",
    "bloom_count_bytes": "bloom_count_kernel<8, HashBytes>",
    "bloom_bin_bytes": "bloom_bin_kernel<8, HashBytes>",
    "bloom_fill": "bloom_fill_kernel",
    "bloom_build_atomic_u64": "bloom_build_kernel<8>",
    "bloom_build_atomic_bytes": "bloom_build_bytes_kernel",
}
SHOW = ("scan_kernel", "popcount_seg", "bytes_eq_kernel", "bloom_", "finalize_packs")
traffic = defaultdict(dict)
for sub in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(sub):
        continue
    for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("\n## PMC pass %s (per-dispatch mean over dispatches)" % os.path.basename(sub))
        for k, ctrs in agg.items():
            if not any(x in k for x in SHOW):
                continue
            print(k)
            for c, vals in sorted(ctrs.items()):
                print("    %-24s n=%-4d mean=%.6g" % (c, len(vals), sum(vals) / len(vals)))
                if c in ("FETCH_SIZE", "WRITE_SIZE"):
                    for label, sub_name in TRAFFIC_KERNELS.items():
                        if sub_name in k:
                            use = vals
                            if any(sp in k for sp in SPLIT):   # full-size dispatches only
                                use = [x for x in vals if x >= max(vals) / 2]
                                print("    %-24s n=%-4d mean=%.6g   (full-size dispatches)" % (c, len(use), sum(use) / len(use)))
                            traffic[label][c] = sum(use) / len(use)

import json
import time
res = {}
for label, t in traffic.items():
    if "FETCH_SIZE" not in t or "WRITE_SIZE" not in t:
        continue
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
    # bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming
    # stores.  bytes_eq_kernel mixes 8 B/lane pointer loads with 16 B/lane payload loads and the bloom builds write through
    # atomics: for those the x2 figure is the guide's rule applied outside its calibrated pattern -- the raw counters are
    # kept beside it.
    fetch_b = t["FETCH_SIZE"] * 1024 * 2
    write_b = t["WRITE_SIZE"] * 1024
    res[label] = {"kernel": TRAFFIC_KERNELS[label], "FETCH_SIZE_KiB_raw": t["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": t["WRITE_SIZE"],
                  "fetch_bytes_corrected_x2": fetch_b, "write_bytes": write_b, "hbm_bytes_per_launch": fetch_b + write_b}
if "headline" in res:
    out = dict(res["headline"])
    out.update({"kernel": "knox::scan_kernel<8,1,ISUM> (headline)", "source": os.path.basename(root),
                "measured": "%s, %s" % (os.path.basename(root).replace("prof_", "round ").replace("round r0", "round "), time.strftime("%Y-%m-%d")),
                "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section)",
                "kernels": {k: v for k, v in res.items() if k != "headline"}})
    print("\n## HBM traffic per launch (PMC FETCH_SIZE x 2 + WRITE_SIZE)\n" + json.dumps(out))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
