#!/bin/bash
# PMC passes over tools/bench_head.py (the one-pass head kernel): usage tools/pmc_head.sh <outdir> [fused|fused32]
# counters in separate passes; --pmc only with --kernel-trace (gpurun rule)
out=$1; which=${2:-fused}
mkdir -p $out
set -e
R=$(cd "$(dirname "$0")/.." && pwd)   # the repo root, from where this script lies
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"
P3="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_INSTS_MFMA"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
n=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  n=$((n+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/p$n -- python3 $R/tools/bench_head.py --only $which --reps 3 > $out/p$n.log 2>&1
done
python3 $R/tools/pmc_summarize.py $out/p1 $out/p2 $out/p3 $out/p4 $out/p5 | grep -i "head_fused"
