#!/bin/bash
# rocprofv3 evidence for bench.py's default workload (run on the GPU box from the repo root):
#   tools/profile_bench.sh <outdir>
# 1. --kernel-trace --stats of a 10-step bench run (per-kernel average durations; the bench line of the same run is kept)
# 2. PMC passes for HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE runs (TCC slots), --pmc only with --kernel-trace
out=$(readlink -f $1); mkdir -p $out
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)   # the repo root, from where this script lies
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-trajectory --no-board-sampler > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_bench_steps10.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 2 --no-graph --no-prof --no-cpu-baseline --no-full-trajectory --no-board-sampler > $out/pmc_$c.log 2>&1
done
cd $R
python3 tools/pmc_summarize.py --json cifar10_unet_b1024_T1000 profiles/r06/pmc_hbm_traffic.txt $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/pmc_hbm_traffic.txt
cp profiles/pmc_traffic.json $out/pmc_traffic.json
rm -rf $out/stats $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
