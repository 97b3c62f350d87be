import csv, glob, sys, collections
root = sys.argv[1]
for p in ("p1", "p2", "p3", "p4"):
    ms = [float(l.split()[2]) for l in open("%s/%s.txt" % (root, p)) if l.startswith("COL")]
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (root, p), recursive=True)
    if not f: continue
    rows = [r for r in csv.DictReader(open(f[0])) if "scan_kernel" in r["Kernel_Name"]]
    by = collections.defaultdict(list)   # counter -> list per dispatch order
    disp = collections.OrderedDict()
    for r in rows:
        disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    d = list(disp.values())
    ncol = len(ms); per = len(d) // ncol
    names = sorted(d[0].keys())
    print("== %s  (per-column mean over %d dispatches)" % (p, per))
    print("col  kernel_ms  " + "  ".join(names))
    for c in range(ncol):
        chunk = d[c * per + per // 2:(c + 1) * per]
        print("%3d  %8.4f  " % (c, ms[c]) + "  ".join("%14.4g" % (sum(x[nm] for x in chunk) / len(chunk)) for nm in names))
