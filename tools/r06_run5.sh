#!/bin/bash
O=gpurun_out/r06_run5
mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -s -k "whole_image" > $O/pytest_img.log 2>&1; tail -8 $O/pytest_img.log
DEFER=dlpm_amd/lib/libdlpm_amd_9802a46f.so
for cin in 32 64 96; do for B in 16 256; do
python tools/bench_resblock_img.py --cin $cin --batch $B 2>&1 | grep -v amdgpu.ids
DLPM_LIB=$DEFER python tools/bench_resblock_img.py --cin $cin --batch $B 2>&1 | grep -v amdgpu.ids | sed 's/^/[deferred counters] /'
done; done | tee $O/resblock_img_phases.txt
