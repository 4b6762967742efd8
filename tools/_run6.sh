mkdir -p gpurun_out/r4h
for i in 1 2 3; do
DLPM_WINO_SPEC=0 python tools/bench_conv.py --gen f4 --reps 20 > gpurun_out/r4h/conv_generic_$i.txt 2>&1
python tools/bench_conv.py --gen f4 --reps 20 > gpurun_out/r4h/conv_spec_$i.txt 2>&1
done
grep -h "H32 128->128\|sum" gpurun_out/r4h/conv_*.txt
python -m pytest tests/test_gpu_sampler.py -m gpu -q -x -k "time_table or graph or teacher or generation_manager" > gpurun_out/r4h/sampler.log 2>&1; tail -3 gpurun_out/r4h/sampler.log
for i in 1 2; do
python bench.py --workload mnist_unet_b256_T1000 --no-cpu-baseline > gpurun_out/r4h/mnist_fused_$i.json 2>&1
done
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/r4h/mnist_*.json')):
    j=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, j['ms_per_step'], j['value'])
"
bash tools/_run5.sh
