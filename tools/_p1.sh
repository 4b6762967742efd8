mkdir -p gpurun_out/p1
python tools/power_probe.py -- python tools/bench_conv.py --gen f4 --only 0 --reps 6000 > gpurun_out/p1/power_wino4_h32_128.txt 2>&1
python tools/power_probe.py -- python tools/bench_conv.py --gen f4 --only 0 --reps 6000 --zeros > gpurun_out/p1/power_wino4_h32_128_zeros.txt 2>&1
python tools/power_probe.py -- python tools/bench_head.py --only fused --reps 40000 > gpurun_out/p1/power_head_fused.txt 2>&1
python tools/power_probe.py -- python bench.py --no-cpu-baseline --no-prof --no-full-trajectory --steps 400 > gpurun_out/p1/power_cifar_step.txt 2>&1
grep -h "steady\|^# python" gpurun_out/p1/*.txt
