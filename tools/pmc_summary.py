"""Mean per dispatch of every counter rocprofv3 --pmc collected under gpurun_out/prof_<tag>/pmc_*/ (tools/pmc_any.sh), per kernel:
    python tools/pmc_summary.py <tag> [kernel-name substring]"""
import collections, csv, glob, json, sys
tag = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "bddmma"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/prof_{tag}/pmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if pat in k:
        print(k[:120], json.dumps({c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())}), f"dispatches={len(next(iter(d.values())))}")
