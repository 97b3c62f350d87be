import csv,glob,sys
from collections import defaultdict
for sub in sys.argv[1:]:
    for f in glob.glob(sub+"/**/*counter_collection.csv", recursive=True):
        agg=defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,c in agg.items():
            if "bytes" in k or "bloom" in k:
                print(k, {n:(len(v), sum(v)/len(v)) for n,v in c.items()})
