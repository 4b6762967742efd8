#!/bin/bash
O=gpurun_out/r06_run7
mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -s -k "whole_image" > $O/pytest_img.log 2>&1; tail -12 $O/pytest_img.log
python -m pytest tests/test_gpu_models.py -m gpu -x -q > $O/pytest_models.log 2>&1; tail -3 $O/pytest_models.log
python -m pytest tests/test_gpu_sampler.py -m gpu -x -q -k "mnist or C2" > $O/pytest_sampler_mnist.log 2>&1; tail -3 $O/pytest_sampler_mnist.log
DEFER=dlpm_amd/lib/libdlpm_amd_9802a46f.so
for cin in 64 128; do for B in 16 256; do
python tools/bench_resblock_img.py --h16 --cin $cin --batch $B 2>&1 | grep -v amdgpu.ids
DLPM_LIB=$DEFER python tools/bench_resblock_img.py --h16 --cin $cin --batch $B 2>&1 | grep -v amdgpu.ids | sed 's/^/[deferred counters] /'
done; done | tee $O/resblock_img16_phases.txt
for i in 1 2; do
for v in 1 0; do
DLPM_RES_IMG16=$v python bench.py --workload mnist_unet_b256_T1000 --no-cpu-baseline --no-full-trajectory --no-board-sampler --steps 300 > $O/bench_mnist_res16_${v}_$i.json 2> $O/bench_mnist_res16_${v}_$i.err
done; done
python tools/prof_layers.py --workload mnist --batch 256 > $O/layers_mnist_b256.txt 2>&1
head -16 $O/layers_mnist_b256.txt; tail -1 $O/layers_mnist_b256.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run7/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], j['ms_per_step'], j['value'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
