"""Copy the rocprofv3 summaries of gpurun_out/prof_<tag> into profiles/<tag>/ (tracked) and derive the per-launch
HBM traffic of the sweep kernels from the FETCH_SIZE / WRITE_SIZE passes.

gfx950 corrections (MI355X_MICROARCH.md §HBM, calibrated here on the plain backward sweep whose byte count is
known: 10 M node words + 5 M {lo,hi} pairs read = 80.0 MB vs FETCH_SIZE 39 104 KiB; 9.5 M potentials written =
38.0 MB vs WRITE_SIZE 37 171 KiB): HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
dst = f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)
shutil.copy(f"{src}/trace/trace_kernel_stats.csv", f"{dst}/kernel_stats.csv")
shutil.copy(f"{src}/bench_under_trace.json", f"{dst}/bench_under_trace.json")
# [sum, count] of a counter over the dispatches of one kernel (tools/profile.sh aggregates on the GPU box)
agg = collections.defaultdict(dict)
for f in sorted(glob.glob(f"{src}/pmc_*.agg.json")):
    for k, d in json.load(open(f)).items():
        for c, (s, n) in d.items():
            agg[k][c] = [s / n] * int(n)   # the mean, repeated: the code below only takes means and counts
with open(f"{dst}/pmc_summary.txt", "w") as out:
    out.write("# rocprofv3 --pmc, one pass per counter group (tools/profile.sh); mean per dispatch of `python bench.py --steps 10`\n")
    for k, d in sorted(agg.items()):
        if "bddmma" in k:
            out.write(f"{k}: " + json.dumps({c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())}) + f"  dispatches={len(next(iter(d.values())))}\n")
traffic = {}
for k, d in agg.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f, w = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        traffic[k] = {"fetch_kib_reported": f, "write_kib_reported": w, "hbm_bytes": (2 * f + w) * 1024}
# the kernel-trace run's average duration of the same kernels, next to their bytes (bench.py: roofline.kernel_us_rocprof)
for r in csv.DictReader(open(f"{dst}/kernel_stats.csv")):
    k = r["Name"].split("(")[0]
    if k in traffic:
        traffic[k]["avg_us_rocprof"] = float(r["AverageNs"]) / 1e3
        traffic[k]["calls_rocprof"] = int(r["Calls"])
# stamp: bench.py quotes these bytes only while the kernel sources are the ones they were measured on
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
traffic["_source_hash"] = bench.source_hash()
json.dump(traffic, open(f"{dst}/traffic.json", "w"), indent=1)
del traffic["_source_hash"]
for k, v in traffic.items():
    if "narrow" in k or "exchange" in k:
        print(k, round(v["hbm_bytes"] / 1e6, 1), "MB")
