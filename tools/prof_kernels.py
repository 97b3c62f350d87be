#!/usr/bin/env python3
"""One table per profiled command (tools/prof_cmd.sh): for every kernel its launches, average duration (rocprofv3
--kernel-trace --stats), the HBM bytes per launch from the PMC passes -- FETCH_SIZE x 2 (the gfx950 correction of
MI355X_MICROARCH.md "HBM": a wide coalesced read is tallied at half its bytes; narrower access shapes are uncalibrated, the
raw KiB are printed beside it) + WRITE_SIZE, both KiB -- the rate those bytes make over the average duration, and the
SQ pass (VALU busy share, wait share, LDS-issue stalls).  usage: prof_kernels.py gpurun_out/prof_<tag>_<name> "<command>" """
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
cmd = sys.argv[2] if len(sys.argv) > 2 else ""


def short(name):
    return name.replace("knox::", "").replace("(anonymous namespace)::", "")[:84]


stats = {}
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]), float(r["Percentage"]))
pmc = defaultdict(lambda: defaultdict(list))
for sub in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))


def mean(k, c):
    v = pmc.get(k, {}).get(c)
    return sum(v) / len(v) if v else None


print("# rocprofv3 of `python3 %s` (%s): kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; one SQ pass" % (cmd, os.path.basename(root)))
print("# hbm_MB = (FETCH_SIZE x 2 + WRITE_SIZE) KiB per launch (x2: gfx950 wide-read correction); TB/s = those bytes / avg duration")
print('# caveat: the x2 is right for 16-byte-per-lane streaming reads (the scans: headline traffic = 1.0003 x algorithmic); kernels that load 4 or 8 bytes\n#   per lane (sort passes, histograms) are counted nearer their raw FETCH_SIZE -- read their hbm_MB as an upper bound; calls of several sizes share a row (averages per launch)')
print("%-84s %6s %11s %6s %9s %9s %9s %6s %6s %6s %7s" % ("kernel", "calls", "avg_us", "pct", "fetchKiB", "writeKiB", "hbm_MB", "TB/s", "valu%", "wait%", "ldsst%"))
for k, (calls, avg, tot, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    fe, wr = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
    hb = (fe * 2 + wr) * 1024 if fe is not None and wr is not None else None
    wc, va, wa, wl = mean(k, "SQ_WAVE_CYCLES"), mean(k, "SQ_ACTIVE_INST_VALU"), mean(k, "SQ_WAIT_ANY"), mean(k, "SQ_WAIT_INST_LDS")
    pc = lambda x: "%6.1f" % (100.0 * x / wc) if (x is not None and wc) else "     -"
    print("%-84s %6d %11.1f %6.2f %9s %9s %9s %6s %s %s %s" % (
        k, calls, avg / 1e3, pct, "%.0f" % fe if fe is not None else "-", "%.0f" % wr if wr is not None else "-",
        "%.1f" % (hb / 1e6) if hb is not None else "-", "%.2f" % (hb / avg / 1e3) if hb is not None else "-", pc(va), pc(wa), pc(wl)))
print("\n## the command's own output (wall clock around whole calls, under the kernel-trace pass)")
