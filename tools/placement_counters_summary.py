#!/usr/bin/env python3
"""Condenses tools/prof_placement.sh: per counter the mean over the 16 fast-pair scans, the 16 slow-pair scans and the 16
read-only launches (dispatch order), and slow / fast."""
import csv
import glob
import os
import sys

out = sys.argv[1]
print("# tools/prof_placement.sh: headline scan kernel, bitset in another placement class (fast) vs in the column's own class (slow)")
print(open(os.path.join(out, "plain.out")).read().strip())
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(os.path.basename(d), "no counter file:", open(d + ".err").read()[-300:])
        continue
    # every pass is a process of its own with a placement of its own: its counters are read against ITS slow / fast time ratio
    own = [l.strip() for l in open(d + ".out") if l.startswith("scan -> fast")]
    print("\n## %s: %s" % (os.path.basename(d), own[0] if own else "(no timing line)"))
    rows = list(csv.DictReader(open(files[0])))
    per = {}
    for r in rows:
        per.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = sorted(per.items())
    scans = [c for (i, k), c in disp if "scan_kernel<8, 1, 1u, false, false, 0u>" in k]
    ros = [c for (i, k), c in disp if "scan_kernel<8, 1, 0u, false, false, 0u>" in k]
    if len(scans) < 32:
        print(os.path.basename(d), "only", len(scans), "scan dispatches")
        continue
    fast, slow = scans[-32:-16], scans[-16:]
    for name in sorted(fast[0]):
        f = sum(c[name] for c in fast) / 16
        s = sum(c[name] for c in slow) / 16
        r = sum(c.get(name, 0.0) for c in ros[-16:]) / max(1, len(ros[-16:]))
        print("%-36s fast %16.1f   slow %16.1f   slow/fast %6.3f   read-only %16.1f" % (name, f, s, s / f if f else float("nan"), r))
