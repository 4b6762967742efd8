mkdir -p gpurun_out/r4b
python -m pytest tests -m gpu -q -s > gpurun_out/r4b/gpu_tests_verbose.log 2>&1
tail -3 gpurun_out/r4b/gpu_tests_verbose.log
for i in 1 2; do
DLPM_LIB=dlpm_amd/lib/libdlpm_amd_r3.so python bench.py --no-cpu-baseline > gpurun_out/r4b/bench_r3_$i.json 2> gpurun_out/r4b/bench_r3_$i.err
python bench.py --no-cpu-baseline > gpurun_out/r4b/bench_new_$i.json 2> gpurun_out/r4b/bench_new_$i.err
done
python bench.py --workload mnist_unet_b256_T1000 --no-cpu-baseline > gpurun_out/r4b/bench_mnist_new.json 2>&1
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, j['ms_per_step'], j['value'], j['roofline']['frac'] if j.get('roofline') else None)
    except Exception as e: print(f, 'ERR', e)
"
