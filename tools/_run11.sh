mkdir -p gpurun_out/r4p
LIBI=dlpm_amd/lib/libdlpm_amd_6c0fb4e7.so
for vs in 0 1 0 1; do
  echo "== DLPM_WINO_VS=$vs" >> gpurun_out/r4p/vsplit_clocks.txt
  DLPM_LIB=$LIBI DLPM_WINO_VS=$vs PHASE_ONLY="3x3 wino" PHASE_FORCE=8 python tools/phase_conv.py 2>&1 | grep -v amdgpu >> gpurun_out/r4p/vsplit_clocks.txt
done
cat gpurun_out/r4p/vsplit_clocks.txt
python -m pytest tests/test_gpu_sampler.py -m gpu -q -s -k "bounded and cifar" > gpurun_out/r4p/bounded_cifar.log 2>&1; grep "post-processed\|passed\|failed" gpurun_out/r4p/bounded_cifar.log | sed 's/GenerationManager-post-processed pixels max |hip - reference| =/ERR/'
