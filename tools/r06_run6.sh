#!/bin/bash
# round 6: n-tile width by declared batch -- parity + the CIFAR net at B = 64 / 256
O=gpurun_out/r06_run6
mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "winograd_f4" > $O/pytest_f4.log 2>&1; tail -3 $O/pytest_f4.log
python -m pytest tests/test_gpu_sampler.py -m gpu -x -q -k "shard_size or full_size or c4_whole" > $O/pytest_shards.log 2>&1; tail -3 $O/pytest_shards.log
python -m pytest tests/test_image_dump.py tests/test_gpu_models.py -m gpu -x -q > $O/pytest_dump_models.log 2>&1; tail -3 $O/pytest_dump_models.log
for B in 64 256; do
for i in 1 2; do
DLPM_WINO4_NQ=128 python bench.py --no-cpu-baseline --no-full-trajectory --no-board-sampler --steps 60 --batch $B > $O/bench_cifar_b${B}_nq128_$i.json 2> $O/bench_cifar_b${B}_nq128_$i.err
python bench.py --no-cpu-baseline --no-full-trajectory --no-board-sampler --steps 60 --batch $B > $O/bench_cifar_b${B}_auto_$i.json 2> $O/bench_cifar_b${B}_auto_$i.err
done
DLPM_WINO4_NQ=64 python bench.py --no-cpu-baseline --no-full-trajectory --no-board-sampler --steps 60 --batch $B > $O/bench_cifar_b${B}_nq64_1.json 2> $O/bench_cifar_b${B}_nq64_1.err
done
python bench.py --no-cpu-baseline --no-full-trajectory --no-board-sampler --steps 60 --batch 64 --dispatch-batch 1024 > $O/bench_cifar_b64_declared1024.json 2> $O/bench_cifar_b64_declared1024.err
python tools/prof_layers.py --workload cifar10 --batch 64 > $O/layers_cifar_b64.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run6/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        k=j['ms_per_step_by_kernel_class']
        print(f.split('/')[-1], j['ms_per_step'], j['value'], {a:k[a] for a in k if 'conv3x3' in a})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
head -30 $O/layers_cifar_b64.txt
