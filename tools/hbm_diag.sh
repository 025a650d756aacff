#!/bin/bash
# usage (GPU box): bash tools/hbm_diag.sh <tag> [hbm_diag.py args]  -> gpurun_out/hbm_diag_<tag>.txt: per kernel, mean per dispatch of each counter
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/hbm_diag_$tag
mkdir -p $out
cd $R
python tools/hbm_diag.py "$@" > $out/times.txt 2>/dev/null
i=0
while read -r pmc; do
  [ -z "$pmc" ] && continue
  i=$((i+1))
  # (a counter set the hardware cannot collect makes rocprofv3 abort and then hang in its signal handler: never without a timeout)
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $out/p$i -o pmc -- python tools/hbm_diag.py "$@" > /dev/null 2> $out/p$i.err
done <<'PMC'
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_sum
TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum
TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_IB_STALL_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum
FETCH_SIZE WRITE_SIZE
PMC
python - $out <<'PY'
import collections, csv, glob, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void bddmma::", "")
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(d + ".txt", "w") as out:
    out.write(open(d + "/times.txt").read())
    for k in sorted(agg):
        if not any(s in k for s in ("narrow", "exchange", "stream")): continue
        out.write(k + "\n")
        for c in sorted(agg[k]):
            s, n = agg[k][c]
            out.write(f"    {c:45s} {s / n:16.1f}   (dispatches {n})\n")
print(open(d + ".txt").read())
PY
rm -rf $out
