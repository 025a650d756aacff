#!/bin/bash
# usage (GPU box): bash tools/kstats.sh <tag> <python script + args...>  -> rocprofv3 kernel-trace stats of one script, top kernels printed
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python "$@" > $out/stdout_trace.txt 2> $out/trace.err
find $out -name "*kernel_trace.csv" -delete
python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(r["Name"][:84].ljust(84), r["Calls"].rjust(7), f'{int(r["TotalDurationNs"]) / 1e6:10.2f} ms', f'{float(r["AverageNs"]) / 1e3:9.2f} us', r["Percentage"])
PY
cat $out/stdout_trace.txt | tail -3
