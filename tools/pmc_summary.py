"""Sums rocprofv3 --pmc counter_collection CSV rows per kernel and counter (average per dispatch)."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    n = len(cnt[k])
    print(f"{k} dispatches={n} " + " ".join(f"{c}={v / n:.4g}" for c, v in sorted(acc[k].items())))
