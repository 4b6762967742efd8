mkdir -p gpurun_out/r4e
python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "fused_small" > gpurun_out/r4e/blocks.log 2>&1; tail -30 gpurun_out/r4e/blocks.log
python -m pytest tests/test_gpu_models.py -m gpu -q -x > gpurun_out/r4e/models.log 2>&1; tail -15 gpurun_out/r4e/models.log
for i in 1 2; do
DLPM_NO_FUSED_BLOCKS=1 python bench.py --workload mnist_unet_b256_T1000 --no-cpu-baseline > gpurun_out/r4e/mnist_unfused_$i.json 2>&1
python bench.py --workload mnist_unet_b256_T1000 --no-cpu-baseline > gpurun_out/r4e/mnist_fused_$i.json 2>&1
done
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/r4e/mnist_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, j['ms_per_step'], j['value']); print(j['ms_per_step_by_kernel_class'])
    except Exception as e: print(f, 'ERR', e, open(f).read()[-1500:])
"
