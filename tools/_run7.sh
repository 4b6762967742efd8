mkdir -p gpurun_out/r4j
python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "head_conv" > gpurun_out/r4j/head.log 2>&1; tail -25 gpurun_out/r4j/head.log
python -m pytest tests/test_gpu_models.py -m gpu -q -x > gpurun_out/r4j/models.log 2>&1; tail -8 gpurun_out/r4j/models.log
for i in 1 2; do
DLPM_NO_HEAD_FUSED=1 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/r4j/cifar_headgemm_$i.json 2>&1
python bench.py --no-cpu-baseline --steps 30 > gpurun_out/r4j/cifar_headfused_$i.json 2>&1
done
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/r4j/cifar_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, j['ms_per_step'], j['value'], j['update_kernel']['frac'], j['update_kernel']['avg_launch_ms'], j['update_kernel']['kernel'][:30]); print({k:v for k,v in j['ms_per_step_by_kernel_class'].items() if 'head' in k or 'igemm' in k})
    except Exception as e: print(f, 'ERR', e, open(f).read()[-1500:])
"
