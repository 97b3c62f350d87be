#!/usr/bin/env python3
"""Static instruction mix of one kernel from the hipcc -S output (whole kernel body)."""
import collections, re, subprocess, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = re.findall(r"^(_ZN4knox\w+):", s, flags=re.M)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for n, d in zip(names, dem):
    if pat in d.replace("knox::", ""):
        i = s.index(n + ":")
        j = s.index(".Lfunc_end", i)
        lines = [l.strip() for l in s[i:j].split("\n") if l.strip() and not l.strip().startswith((";", "."))][1:]
        c = collections.Counter(l.split()[0] for l in lines)
        f64 = sum(v for k, v in c.items() if "f64" in k)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print("%s\n  total %d  VALU %d (f64 %d)  SALU %d  vmem %d  lds %d  nop %d" % (
            d.replace("knox::", ""), len(lines), valu, f64, sum(v for k, v in c.items() if k.startswith("s_") and k not in ("s_nop", "s_waitcnt")),
            sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_"))), sum(v for k, v in c.items() if k.startswith("ds_")), c.get("s_nop", 0)))
        if len(sys.argv) > 3:
            for k, v in c.most_common(int(sys.argv[3])):
                print("    %-28s %d" % (k, v))
