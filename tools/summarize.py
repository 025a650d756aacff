import csv, glob, collections, sys
d = sys.argv[1]
print(open(f"{d}/trace/trace_kernel_stats.csv").read()[:1500])
for f in sorted(glob.glob(f"{d}/pmc_*/pmc_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in agg.items():
        if "bddmma" in k and ("narrow" in k or "exchange" in k or "gather" in k):
            print(k[:50], {c: round(sum(v)/len(v),1) for c, v in dd.items()}, "n", len(next(iter(dd.values()))))
