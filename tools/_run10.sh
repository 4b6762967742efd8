mkdir -p gpurun_out/r4o
python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "winograd_f4" > gpurun_out/r4o/kernels_f4_vs.log 2>&1; grep "err\|passed\|failed" gpurun_out/r4o/kernels_f4_vs.log | tail -14
for i in 1 2 3; do
DLPM_WINO_VS=0 python tools/bench_conv.py --gen f4 --reps 20 > gpurun_out/r4o/conv_vs0_$i.txt 2>&1
python tools/bench_conv.py --gen f4 --reps 20 > gpurun_out/r4o/conv_vs1_$i.txt 2>&1
done
for f in gpurun_out/r4o/conv_vs*_1.txt; do echo $f; grep -v amdgpu $f; done
grep -h "^sum" gpurun_out/r4o/conv_vs0_*.txt; echo; grep -h "^sum" gpurun_out/r4o/conv_vs1_*.txt
python tools/err_report.py 2>&1 | grep -v amdgpu
