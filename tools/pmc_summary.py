"""Summarise rocprofv3 --pmc counter_collection CSVs: mean per dispatch of each counter, per kernel."""
import csv, sys, collections, glob
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(path.split("/")[-1], k)
        for c, v in cs.items():
            print(f"   {c:28s} mean={sum(v)/len(v):16.1f}  n={len(v)}")
