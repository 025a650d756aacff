#!/bin/bash
# usage (GPU box): bash tools/pmc_any.sh <tag> <python script + args...>  -> kernel trace stats + SQ / LDS / traffic counters under gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python "$@" > $out/stdout_trace.txt 2> $out/trace.err
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pmc --output-format csv -d $out/pmc_$name -o pmc -- python "$@" > /dev/null 2> $out/pmc_$name.err
done
ls $out
