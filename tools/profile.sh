#!/bin/bash
# Usage (on the GPU box, via gpurun): bash tools/profile.sh <tag> [bench args...]
# Writes rocprofv3 kernel-trace stats and PMC passes under gpurun_out/prof_<tag>/.  Every pass runs under `timeout` (a --pmc pass was seen to hang for an
# hour in round 6 with the configs[0] / configs[1] legs of bench.py in it: the passes profile the headline loop only, --no-configs).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-lbfgs --no-configs "$@" > $out/bench_under_trace.json 2> $out/trace.err
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $out/pmc_$name -o pmc -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-lbfgs --no-configs "$@" > /dev/null 2> $out/pmc_$name.err
  # per-dispatch rows -> per-kernel sums on the box (the raw CSVs of three workloads exceed the 64 MiB that travel back)
  python - $out/pmc_$name <<'PY'
import collections, csv, glob, json, shutil, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
shutil.rmtree(d)
json.dump(agg, open(d + ".agg.json", "w"))
PY
done
# the per-dispatch trace is not needed for the summaries and would push gpurun_out/ over the 64 MiB that travel back
find $out -name "*kernel_trace.csv" -delete
find $out -name "*.csv" | head -30
