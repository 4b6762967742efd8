#!/bin/bash
# round 6, GPU call 2: the whole-image F(4x4) kernel (32 channels, 32x32) -- parity, A/B on the MNIST step, phase cycles; F4_RES_AHEAD A/B
O=gpurun_out/r06_run2
mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "whole_image or winograd_f4" > $O/pytest_f4.log 2>&1; tail -5 $O/pytest_f4.log
python -m pytest tests/test_gpu_models.py -m gpu -x -q > $O/pytest_models.log 2>&1; tail -3 $O/pytest_models.log
for i in 1 2; do
for v in 1 0; do
DLPM_WINO4_IMG=$v python bench.py --workload mnist_unet_b256_T1000 --no-cpu-baseline --no-full-trajectory --no-board-sampler --steps 300 > $O/bench_mnist_img${v}_$i.json 2> $O/bench_mnist_img${v}_$i.err
done; done
DEFER=dlpm_amd/lib/libdlpm_amd_9802a46f.so
for B in 16 64 256; do DLPM_LIB=$DEFER PHASE_ONLY="mnist H32 32->32" PHASE_B=$B PHASE_FORCE=8 PHASE_REPS=50 python tools/phase_conv.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[img, deferred] /"; done > $O/phase_img.txt
for B in 16 64 256; do PHASE_ONLY="mnist H32 32->32" PHASE_B=$B PHASE_FORCE=8 PHASE_REPS=50 python tools/phase_conv.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[img, product ] /"; done >> $O/phase_img.txt
for B in 16 64 256; do DLPM_WINO4_IMG=0 PHASE_ONLY="mnist H32 32->32" PHASE_B=$B PHASE_FORCE=8 PHASE_REPS=50 python tools/phase_conv.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[2+2,  product ] /"; done >> $O/phase_img.txt
cat $O/phase_img.txt
python tools/prof_layers.py --workload mnist --batch 256 > $O/layers_mnist_b256_img.txt 2>&1
RA=dlpm_amd/lib/libdlpm_amd_$(python -c "import hashlib;print(hashlib.sha256(b'F4_RES_AHEAD=1').hexdigest()[:8])").so
ls -la $RA
for i in 1 2 3; do
python tools/bench_conv.py --gen f4 --reps 20 2>&1 | grep -v amdgpu.ids | sed "s/^/[product   $i] /"
DLPM_LIB=$RA python tools/bench_conv.py --gen f4 --reps 20 2>&1 | grep -v amdgpu.ids | sed "s/^/[res-ahead $i] /"
done > $O/conv_layers_res_ahead.txt
cat $O/conv_layers_res_ahead.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run2/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], j['ms_per_step'], j['value'], {k:v for k,v in j['ms_per_step_by_kernel_class'].items() if 'conv3x3' in k})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-500:])
PY
