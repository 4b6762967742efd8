#!/bin/bash
# tools/validate_build.sh <tag>: on the GPU box, from the repo root -- the full GPU suite (verbose), the driver's bench line, the MNIST and CelebA-64
# workloads, rocprofv3 kernel stats + the PMC traffic record of THIS build (profiles/pmc_traffic.json), and the bench line again with `traffic` filled in.
tag=${1:-r6}
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -q -s > gpurun_out/$tag/gpu_tests_verbose.log 2>&1
tail -3 gpurun_out/$tag/gpu_tests_verbose.log
python __graft_entry__.py smoke > gpurun_out/$tag/smoke.log 2>&1; tail -1 gpurun_out/$tag/smoke.log
python bench.py > gpurun_out/$tag/bench_cifar.json 2> gpurun_out/$tag/bench_cifar.err
python bench.py --workload mnist_unet_b256_T1000 --no-cpu-baseline > gpurun_out/$tag/bench_mnist.json 2>&1
python bench.py --workload celeba64_unet_b256_T1000 --no-cpu-baseline --steps 20 > gpurun_out/$tag/bench_celeba64_shard.json 2>&1
python bench.py --no-cpu-baseline --no-full-trajectory --clip > gpurun_out/$tag/bench_cifar_clip.json 2>&1
python bench.py --no-cpu-baseline --no-full-trajectory --batch 64 --steps 60 > gpurun_out/$tag/bench_cifar_b64.json 2>&1
bash tools/profile_bench.sh gpurun_out/$tag/prof > gpurun_out/$tag/profile_bench.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/$tag/pmc_traffic.json
cp profiles/r06/pmc_hbm_traffic.txt gpurun_out/$tag/pmc_hbm_traffic.txt 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/$tag/bench_cifar_with_traffic.json 2>&1
python tools/prof_layers.py --workload cifar10 --batch 1024 > gpurun_out/$tag/layers_cifar_b1024.txt 2>&1
python tools/prof_layers.py --workload mnist --batch 256 > gpurun_out/$tag/layers_mnist_b256.txt 2>&1
python -c "
import json
for f in ['bench_cifar','bench_mnist','bench_celeba64_shard','bench_cifar_clip','bench_cifar_b64','bench_cifar_with_traffic']:
    try:
        j=json.loads([l for l in open('gpurun_out/$tag/%s.json'%f) if l.startswith('{')][-1]); print(f, j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['traffic'], (j['update_kernel'] or {}).get('frac'), (j['update_kernel'] or {}).get('traffic'), (j.get('cpu_baseline') or {}).get('value'), j.get('board_power_w'), j.get('sclk_mhz'), j.get('package_limit_w'))
    except Exception as e: print(f, 'ERR', e)
"
