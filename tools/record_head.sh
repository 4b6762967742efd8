#!/bin/bash
# tools/record_head.sh <outdir>: the one-pass head kernel's records of one build, on the GPU box from the repo root --
# the access-pattern microbenchmark, the head's forms side by side, the timing-only ablations (HF_ABLATE variant libraries built
# beforehand with DLPM_BUILD_DEFS="HF_ABLATE=<bits>" python -m dlpm_amd.build), the phase clocks (DLPM_PHASE_TIMING library), the clock the
# chip holds (DLPM_BUILD_DEFS="DLPM_PHASE_TIMING HF_CLOCK_ONLY [HF_ABLATE=1|2|3|7|8]") and the SQ / FETCH / WRITE counters.
out=${1:-gpurun_out/head_forms}
mkdir -p $out
# the access-pattern microbenchmark is built on demand (no binary in the tree)
[ -x tools/mb/stream_pattern ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/mb/stream_pattern tools/mb/stream_pattern.hip
./tools/mb/stream_pattern > $out/mb_stream_pattern.txt 2>&1
{
  echo "# tools/bench_head.py --reps 30: the one-pass kernel on the bf16 pipe (fused) and on the fp32 MFMA (fused32), round 3's GEMM + gather pair, the VALU kernel"
  python tools/bench_head.py --reps 30 2>/dev/null
  echo "# again, one-pass forms only"
  python tools/bench_head.py --reps 30 --only fused 2>/dev/null | grep total
  python tools/bench_head.py --reps 30 --only fused32 2>/dev/null | grep total
  echo "# CelebA-64 head [256,64,64,128] -> 3"
  python tools/bench_head.py --reps 30 --B 256 --H 64 --only fused 2>/dev/null | grep total
  python tools/bench_head.py --reps 30 --B 256 --H 64 --only pair 2>/dev/null | grep total
} > $out/bench_head_forms.txt
{
  echo "# timing-only ablations of k_head_fused (HF_ABLATE bits: 1 no MFMAs, 2 no SiLU / split arithmetic, 4 no gather, 8 no HBM reloads); results are wrong by construction"
  for a in 0 1 2 3 4 8 11; do
    if [ $a = 0 ]; then lib=dlpm_amd/lib/libdlpm_amd.so; else
      tag=$(python -c "import hashlib; print(hashlib.sha256('HF_ABLATE=$a'.encode()).hexdigest()[:8])"); lib=dlpm_amd/lib/libdlpm_amd_$tag.so; fi
    [ -f $lib ] || continue
    echo "HF_ABLATE=$a  $(DLPM_LIB=$lib python tools/bench_head.py --reps 30 --only fused 2>/dev/null | grep total)"
  done
} > $out/ablations.txt
tag=$(python -c "import hashlib; print(hashlib.sha256('DLPM_PHASE_TIMING'.encode()).hexdigest()[:8])")
[ -f dlpm_amd/lib/libdlpm_amd_$tag.so ] && DLPM_LIB=dlpm_amd/lib/libdlpm_amd_$tag.so python tools/bench_head.py --reps 30 --only fused > $out/phase_clocks.txt 2>/dev/null
{
  echo "# the clock the chip holds inside k_head_fused (shader cycles / 100-MHz ticks over each workgroup's life; DLPM_PHASE_TIMING + HF_CLOCK_ONLY"
  echo "# libraries: two atomics per workgroup): the full kernel, no MFMAs (1), no SiLU / split arithmetic (2), memory only (3), memory only"
  echo "# without the gather (7), arithmetic only (8).  cycles = ms x MHz."
  for a in 0 1 2 3 7 8; do
    defs="DLPM_PHASE_TIMING HF_CLOCK_ONLY"; [ $a != 0 ] && defs="$defs HF_ABLATE=$a"
    tag=$(python -c "import hashlib,sys; print(hashlib.sha256(' '.join(sorted(sys.argv[1].split())).encode()).hexdigest()[:8])" "$defs")
    [ -f dlpm_amd/lib/libdlpm_amd_$tag.so ] || continue
    echo "HF_ABLATE=$a  $(DLPM_LIB=dlpm_amd/lib/libdlpm_amd_$tag.so python tools/bench_head.py --reps 30 --only fused 2>/dev/null | grep -E -o 'clock [0-9]+ MHz|total [0-9.]+ ms')" | tr '\n' ' '; echo
  done
} > $out/held_clock.txt
cat $out/held_clock.txt
bash tools/pmc_head.sh $GRAFT_REPO_ROOT/$out/pmc fused > $out/pmc_head_fused.txt 2>&1
rm -rf $out/pmc
cat $out/ablations.txt $out/phase_clocks.txt
