# round 4, VERDICT items 4a and 5: held clock per ablation (F(4x4) kernel), held clock + cycle-level MFMA busy of k_conv_split
mkdir -p gpurun_out/r4g
LIBI=dlpm_amd/lib/libdlpm_amd_6c0fb4e7.so
for abl in 0 64 16 80 3 32 87; do
  echo "== DLPM_WABL=$abl" >> gpurun_out/r4g/wino4_ablation_clocks.txt
  DLPM_LIB=$LIBI DLPM_WABL=$abl PHASE_ONLY="3x3 wino" PHASE_FORCE=8 python tools/phase_conv.py >> gpurun_out/r4g/wino4_ablation_clocks.txt 2>&1
done
echo "== k_conv_split (bf16x3), DLPM_PHASE_TIMING build" > gpurun_out/r4g/split_clocks.txt
DLPM_LIB=$LIBI PHASE_ONLY="1x1" PHASE_FORCE=16 python tools/phase_conv.py >> gpurun_out/r4g/split_clocks.txt 2>&1
for idx in 3 4 5; do
  bash tools/pmc_gemm.sh $PWD/gpurun_out/r4g/pmc_split_$idx $idx > gpurun_out/r4g/pmc_split_$idx.txt 2>&1
done
rm -rf gpurun_out/r4g/pmc_split_*/p*/  2>/dev/null
cat gpurun_out/r4g/wino4_ablation_clocks.txt gpurun_out/r4g/split_clocks.txt | grep -v amdgpu.ids
grep -h "SQ_BUSY_CYCLES\|MFMA_BUSY\|SQ_WAVE_CYCLES\|SQ_WAIT_ANY " gpurun_out/r4g/pmc_split_*.txt
