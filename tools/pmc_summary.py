"""Summarise rocprofv3 --pmc CSV passes per kernel (scratch tool)."""
import csv, sys, os, collections, json
d = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for p in sorted(os.listdir(d)):
    f = os.path.join(d, p, p + "_counter_collection.csv")
    if not os.path.exists(f): continue
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in tot:
    if "pt::" not in k: continue
    print(k)
    for c, v in sorted(tot[k].items()):
        print(f"   {c:28s} total {v:16.0f}  per-dispatch {v / max(cnt[k][c],1):14.1f}  dispatches {cnt[k][c]}")
