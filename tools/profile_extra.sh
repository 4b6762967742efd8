#!/bin/bash
out=$(readlink -f gpurun_out/r06_prof_extra); mkdir -p $out
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_mnist -- python3 $R/bench.py --workload mnist_unet_b256_T1000 --steps 50 --warmup 5 --no-cpu-baseline --no-full-trajectory --no-board-sampler > $out/bench_mnist_under_rocprof.json 2> $out/bench_mnist_under_rocprof.err
f=$(find $out/stats_mnist -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_mnist_steps50.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_b64 -- python3 $R/bench.py --batch 64 --steps 30 --warmup 5 --no-cpu-baseline --no-full-trajectory --no-board-sampler > $out/bench_cifar_b64_under_rocprof.json 2> $out/bench_cifar_b64_under_rocprof.err
f=$(find $out/stats_b64 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_cifar_b64_steps30.csv
rm -rf $out/stats_mnist $out/stats_b64
cd $R
head -14 $out/kernel_stats_mnist_steps50.csv | cut -c1-150
head -10 $out/kernel_stats_cifar_b64_steps30.csv | cut -c1-150
