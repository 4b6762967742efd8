#!/bin/bash
# PMC passes over tools/bench_conv.py (one layer shape): usage tools/pmc_conv.sh <outdir> <shape index> [env assignments...]
# counters in separate passes (8 SQ slots per pass); --pmc only with --kernel-trace (gpurun rule)
out=$1; idx=$2; shift 2
mkdir -p $out
export TMPDIR=/tmp
for v in "$@"; do export "$v"; done
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA"
n=0
for P in "$P1" "$P2"; do
  n=$((n+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/p$n -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --only $idx --reps 3 > $out/p$n.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summarize.py $out/p1 $out/p2 | grep -i "wino4\|igemm" 
