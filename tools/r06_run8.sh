#!/bin/bash
# round 6: split-K under a small declared batch -- parity + the CIFAR net at B = 64 / 16
O=gpurun_out/r06_run8
mkdir -p $O
python -m pytest tests/test_gpu_models.py -m gpu -x -q -s -k "declared or forward_matches or batch_tail" > $O/pytest_models.log 2>&1; tail -8 $O/pytest_models.log
python -m pytest tests/test_gpu_sampler.py -m gpu -x -q -k "shard_size or full_size or c4_whole" > $O/pytest_shards.log 2>&1; tail -3 $O/pytest_shards.log
for B in 64 16; do
for i in 1 2; do
DLPM_KSPLIT=0 python bench.py --no-cpu-baseline --no-full-trajectory --no-board-sampler --steps 60 --batch $B > $O/bench_cifar_b${B}_nosplit_$i.json 2> $O/bench_cifar_b${B}_nosplit_$i.err
python bench.py --no-cpu-baseline --no-full-trajectory --no-board-sampler --steps 60 --batch $B > $O/bench_cifar_b${B}_split_$i.json 2> $O/bench_cifar_b${B}_split_$i.err
done; done
python bench.py --no-cpu-baseline --no-full-trajectory --no-board-sampler --steps 40 > $O/bench_cifar_b1024.json 2> $O/bench_cifar_b1024.err
python tools/prof_layers.py --workload cifar10 --batch 64 > $O/layers_cifar_b64.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run8/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        k=j['ms_per_step_by_kernel_class']
        print(f.split('/')[-1], j['ms_per_step'], j['value'], {a:k[a] for a in k if 'conv3x3' in a or 'split' in a or 'groupnorm' in a})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
head -24 $O/layers_cifar_b64.txt
