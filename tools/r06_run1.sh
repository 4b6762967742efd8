#!/bin/bash
# round 6, GPU call 1: the lockstep-epilogue experiment + this round's baselines on one box
O=gpurun_out/r06_run1
mkdir -p $O
bash tools/exp_epilogue_lockstep.sh > $O/epilogue_lockstep.txt 2>&1
for i in 1 2; do
python bench.py --workload mnist_unet_b256_T1000 --no-cpu-baseline --no-full-trajectory --steps 300 > $O/bench_mnist_$i.json 2> $O/bench_mnist_$i.err
done
python bench.py --no-cpu-baseline --steps 40 > $O/bench_cifar.json 2> $O/bench_cifar.err
python bench.py --no-cpu-baseline --no-full-trajectory --steps 40 --clip > $O/bench_cifar_clip.json 2> $O/bench_cifar_clip.err
python bench.py --no-cpu-baseline --no-full-trajectory --steps 40 --deterministic > $O/bench_cifar_dlim.json 2> $O/bench_cifar_dlim.err
python bench.py --no-cpu-baseline --no-full-trajectory --steps 40 --batch 64 > $O/bench_cifar_b64.json 2> $O/bench_cifar_b64.err
python tools/prof_layers.py --workload mnist --batch 256 > $O/layers_mnist_b256.txt 2>&1
python -m pytest tests/test_bench_contract.py tests/test_gpu_kernels.py -m gpu -x -q -k "bench or winograd_f4 or split" > $O/pytest_subset.log 2>&1
tail -3 $O/pytest_subset.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run1/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], j['ms_per_step'], j['value'], 'roof', j['roofline'] and j['roofline']['frac'], 'upd', j['update_kernel'] and j['update_kernel']['frac'], 'unfused', j['update_kernel_unfused'] and (j['update_kernel_unfused']['frac'], j['update_kernel_unfused']['avg_launch_ms']), 'board', j['board_power_w'], j['sclk_mhz'], j['package_limit_w'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -60 $O/epilogue_lockstep.txt
