mkdir -p gpurun_out/r4f
python -m pytest tests -m gpu -q -s > gpurun_out/r4f/gpu_tests_verbose.log 2>&1
tail -4 gpurun_out/r4f/gpu_tests_verbose.log
for i in 1 2; do
DLPM_NO_FUSED_BLOCKS=1 python bench.py --workload mnist_unet_b256_T1000 --no-cpu-baseline > gpurun_out/r4f/mnist_unfused_$i.json 2>&1
python bench.py --workload mnist_unet_b256_T1000 --no-cpu-baseline > gpurun_out/r4f/mnist_fused_$i.json 2>&1
done
python tools/bench_toy.py > gpurun_out/r4f/bench_toy.txt 2>&1
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/r4f/mnist_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, j['ms_per_step'], j['value']); print(j['ms_per_step_by_kernel_class'])
    except Exception as e: print(f, 'ERR', e, open(f).read()[-1500:])
"
cat gpurun_out/r4f/bench_toy.txt
