#!/bin/bash
# PMC passes over tools/bench_gemm.py (one 1x1 shape): usage tools/pmc_gemm.sh <outdir> <shape index>
# counters in separate passes; --pmc only with --kernel-trace (gpurun rule)
out=$1; idx=$2
mkdir -p $out
set -e
R=$(cd "$(dirname "$0")/.." && pwd)   # the repo root, from where this script lies
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
n=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  n=$((n+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/p$n -- python3 $R/tools/bench_gemm.py --only $idx --reps 3 > $out/p$n.log 2>&1
done
python3 $R/tools/pmc_summarize.py $out/p1 $out/p2 $out/p3 $out/p4 $out/p5 | grep -i "split\|igemm"
