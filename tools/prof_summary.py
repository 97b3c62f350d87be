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


print("# rocprofv3 summary of `python3 bench.py --no-cpu-baseline` = the default run, 20 steps + 3 warmup (%s)" % os.path.basename(root))
print("# scan_kernel<8, 1, 1u, false, false, 0u> = the scans: 12 of bench.py's one setup check + 3 warmup + 20 timed + 30 of the 0 % / 100 % selectivity runs;")
print("# <..., 1u> (FLV 1) = the arena's probes (chunk labels, write-buffer verification: alternating read-only / scan launches over 1 GiB samples);")
print("# scan_kernel<8, 1, 0u, ...> = the read-only form (setup check, and the ceiling after the timed region)")
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("\n## kernel stats (--kernel-trace --stats)")
    print("%-92s %7s %12s %12s %8s" % ("kernel", "calls", "avg_ns", "total_ns", "pct"))
    for r in csv.DictReader(open(f)):
        print("%-92s %7s %12s %12s %8s" % (short(r["Name"]), r["Calls"], r["AverageNs"], r["TotalDurationNs"], r["Percentage"]))
tj = os.path.join(root, "trace_bench.json")
if os.path.exists(tj):
    print("\n## bench line under the profiler\n" + open(tj).read().strip())
traffic = {}
for sub in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(sub):
        continue
    for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("\n## PMC pass %s (per-dispatch mean over dispatches)" % os.path.basename(sub))
        for k, ctrs in agg.items():
            if "scan_kernel" not in k and "popcount_seg" not in k:
                continue
            print(k)
            for c, vals in sorted(ctrs.items()):
                print("    %-24s n=%-4d mean=%.6g" % (c, len(vals), sum(vals) / len(vals)))
                if "scan_kernel<8, 1, 1u, false, false, 0u>" in k and c in ("FETCH_SIZE", "WRITE_SIZE"):
                    traffic[c] = sum(vals) / len(vals)

if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly
    # half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for
    # 16-B-per-lane streaming stores.
    fetch_b = traffic["FETCH_SIZE"] * 1024 * 2
    write_b = traffic["WRITE_SIZE"] * 1024
    import json
    t = {"kernel": "knox::scan_kernel<8,1,ISUM> (headline)", "FETCH_SIZE_KiB_raw": traffic["FETCH_SIZE"],
         "WRITE_SIZE_KiB_raw": traffic["WRITE_SIZE"], "fetch_bytes_corrected_x2": fetch_b, "write_bytes": write_b,
         "hbm_bytes_per_launch": fetch_b + write_b, "source": os.path.basename(root),
         "measured": "round 2, %s" % __import__("time").strftime("%Y-%m-%d"),
         "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section)"}
    print("\n## HBM traffic per launch of the headline kernel\n" + json.dumps(t))
    if len(sys.argv) > 2:
        json.dump(t, open(sys.argv[2], "w"), indent=1)
