mkdir -p gpurun_out/r4k
python tools/bench_head.py > gpurun_out/r4k/bench_head.txt 2>&1; cat gpurun_out/r4k/bench_head.txt
export TMPDIR=/tmp; cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS"
n=0
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4k/p$n -- python3 $GRAFT_REPO_ROOT/tools/bench_head.py --only fused --reps 3 > $GRAFT_REPO_ROOT/gpurun_out/r4k/p$n.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summarize.py gpurun_out/r4k/p1 gpurun_out/r4k/p2 gpurun_out/r4k/p3 gpurun_out/r4k/p4 > gpurun_out/r4k/pmc_head_fused.txt 2>&1
grep "head_fused" gpurun_out/r4k/pmc_head_fused.txt
rm -rf gpurun_out/r4k/p1 gpurun_out/r4k/p2 gpurun_out/r4k/p3 gpurun_out/r4k/p4
