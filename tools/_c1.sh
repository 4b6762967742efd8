mkdir -p gpurun_out/c1
run() { # name, extra args
  rm -rf /tmp/gen_$1; t0=$(date +%s.%N)
  python -m dlpm_amd.cli --config cifar10 --generate 5000 --reverse_steps 1000 --synthetic_weights 4321 --set_seed 1 --gen_data_path /tmp/gen_$1 $2 > gpurun_out/c1/$1.log 2>&1
  t1=$(date +%s.%N); w=$(python -c "print('%.1f'%($t1-$t0))")
  n=$(ls /tmp/gen_$1 | wc -l)
  sum=$(cd /tmp/gen_$1 && for i in $(seq 0 1999); do cat $i.png; done | sha256sum | cut -c1-16)
  echo "$1: $n png files, $w s wall -> $(python -c "print('%.2f'%(5000/$w))") images/s end to end (process start to last file), sha256 of files 0..1999: $sum"
}
run default_auto ""
run reference_chunks_64 "--device_batch 0"
