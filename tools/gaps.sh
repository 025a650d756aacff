#!/bin/bash
# usage (GPU box): bash tools/gaps.sh <tag> <python script + args...>  -> idle time between consecutive kernels of a traced run, by the kernel before the gap
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/gaps_$tag
mkdir -p $out
cd $R
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o trace -- python "$@" > $out/stdout_trace.txt 2> $out/trace.err
python - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
rows = rows[len(rows) // 3:]   # the steady part
gap = collections.defaultdict(lambda: [0, 0.0]); busy = 0.0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = max(0, s1 - e0) / 1e3
    key = n0.split("<")[0].split("(")[0][-40:] + " -> " + n1.split("<")[0].split("(")[0][-40:]
    gap[key][0] += 1; gap[key][1] += g
    busy += (e0 - s0) / 1e3
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f"span {span / 1e3:.2f} ms, kernels busy {busy / 1e3:.2f} ms ({100 * busy / span:.1f} %)")
for k, (c, t) in sorted(gap.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{t / 1e3:8.2f} ms total  {t / c:7.2f} us avg  x{c:5d}  {k}")
PY
find $out -name "*kernel_trace.csv" -delete
tail -2 $out/stdout_trace.txt
