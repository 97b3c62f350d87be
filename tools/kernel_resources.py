#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr file) per kernel."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = txt.split("Function Name: ")[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
print("%-100s %5s %5s %7s %4s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS"))
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    d = d.replace("knox::", "").replace("(knox::ScanArgs)", "")
    if pat in d:
        print("%-100s %5d %5d %7d %4d %6d" % (d[:100], g("VGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                                              g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
