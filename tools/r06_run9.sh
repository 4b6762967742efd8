#!/bin/bash
# round 6: persistent workgroups for the 8-wave F(4x4) shape (DLPM_WINO4_PERSIST=1) -- layer times + digests, then the CIFAR step A/B
O=gpurun_out/r06_run9
mkdir -p $O
for i in 1 2 3; do
python tools/bench_conv.py --gen f4 --reps 20 2>&1 | grep -v amdgpu.ids | sed "s/^/[plain      $i] /"
DLPM_WINO4_PERSIST=1 python tools/bench_conv.py --gen f4 --reps 20 2>&1 | grep -v amdgpu.ids | sed "s/^/[persistent $i] /"
done > $O/conv_layers_persistent.txt
cat $O/conv_layers_persistent.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-full-trajectory --no-board-sampler --steps 60 > $O/bench_cifar_plain_$i.json 2> $O/bench_cifar_plain_$i.err
DLPM_WINO4_PERSIST=1 python bench.py --no-cpu-baseline --no-full-trajectory --no-board-sampler --steps 60 > $O/bench_cifar_pers_$i.json 2> $O/bench_cifar_pers_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run9/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['avg_launch_ms'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
