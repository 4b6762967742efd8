#!/bin/bash
# tools/exp_epilogue_lockstep.sh -- VERDICT r05 next #1: why does the F(4x4) register epilogue cost 5.8 k cycles on a 16-workgroup launch and
# 25.8 k on a 256-workgroup one (profiles/r05/mnist_narrow_f4/epilogue_vs_grid.txt)?  Run on the GPU box from the repo root; needs the
# five libraries built HERE beforehand (hipcc cross-compiles):
#     python -m dlpm_amd.build                                                                      product
#     DLPM_BUILD_DEFS="DLPM_PHASE_TIMING" python -m dlpm_amd.build                                  counters, atomics IN PLACE (round 5's form)
#     DLPM_BUILD_DEFS="DLPM_PHASE_TIMING F4N_ABL=7" python -m dlpm_amd.build                        + epilogue without stores / residual / statistics
#     DLPM_BUILD_DEFS="DLPM_PHASE_TIMING DLPM_PHASE_DEFER" python -m dlpm_amd.build                 counters in scalar registers, atomics at the END
#     DLPM_BUILD_DEFS="DLPM_PHASE_TIMING DLPM_PHASE_DEFER F4N_ABL=7" python -m dlpm_amd.build
# Hypothesis under test: the in-place counters ARE the effect -- thread 0 of every workgroup issues a 64-bit atomicAdd onto the same
# cache line at each marker; on gfx9 an atomic without return counts on vmcnt, so the wave's next vmcnt wait (the epilogue's bias /
# residual loads) waits for the atomic, and with every CU in lockstep the atomics queue up at the one address: cost ~ workgroups that
# arrive together.  If so, (1) the deferred build's epilogue does not depend on the grid, (2) neither does the product kernel's launch
# time beyond its work, (3) all-zero operands change nothing.
tag() { python - "$@" <<'EOF'
import hashlib, sys
print(hashlib.sha256(' '.join(sorted(sys.argv[1:])).encode()).hexdigest()[:8])
EOF
}
L=dlpm_amd/lib
INPLACE=$L/libdlpm_amd_$(tag DLPM_PHASE_TIMING).so
INPLACE7=$L/libdlpm_amd_$(tag DLPM_PHASE_TIMING F4N_ABL=7).so
DEFER=$L/libdlpm_amd_$(tag DLPM_PHASE_TIMING DLPM_PHASE_DEFER).so
DEFER7=$L/libdlpm_amd_$(tag DLPM_PHASE_TIMING DLPM_PHASE_DEFER F4N_ABL=7).so
ls -la $INPLACE $INPLACE7 $DEFER $DEFER7 $L/libdlpm_amd.so || exit 1
sweep() {   # label, library ("" = product), shape filter, batches, extra env
    for B in $4; do
        env DLPM_LIB=$2 PHASE_ONLY="$3" PHASE_B=$B PHASE_FORCE=8 PHASE_REPS=50 $5 python tools/phase_conv.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[$1] /"
    done
}
echo "===== A. narrow kernel, H16 64->64 (4 MFMA + 3 helper waves; workgroups = B)"
for rep in 1 2; do
sweep "product           " ""        "mnist H16 64->64" "16 64 128 256 512 1024"
done
sweep "in-place, stripped" $INPLACE7 "mnist H16 64->64" "16 64 128 256 512 1024"
sweep "deferred, stripped" $DEFER7   "mnist H16 64->64" "16 64 128 256 512 1024"
sweep "in-place, full    " $INPLACE  "mnist H16 64->64" "16 64 128 256 512 1024"
sweep "deferred, full    " $DEFER    "mnist H16 64->64" "16 64 128 256 512 1024"
echo "===== A'. the same on ALL-ZERO operands (power vs structure)"
sweep "in-place, stripped, zeros" $INPLACE7 "mnist H16 64->64" "16 256" PHASE_ZEROS=1
sweep "deferred, stripped, zeros" $DEFER7   "mnist H16 64->64" "16 256" PHASE_ZEROS=1
sweep "product, zeros           " ""        "mnist H16 64->64" "16 256" PHASE_ZEROS=1
echo "===== B. narrow kernel, H32 32->32 (2 MFMA + 2 helper waves, two workgroups per CU; workgroups = 4 B)"
sweep "product           " ""        "mnist H32 32->32" "4 16 64 128 256"
sweep "in-place, full    " $INPLACE  "mnist H32 32->32" "4 16 64 128 256"
sweep "deferred, full    " $DEFER    "mnist H32 32->32" "4 16 64 128 256"
echo "===== C. the 8-wave kernel at its worst CIFAR layer, H32 128->128 + residual (workgroups = 4 B: 64, 256, 1024, 4096)"
sweep "product           " ""        "cifar8w" "16 64 256 1024"
sweep "in-place          " $INPLACE  "cifar8w" "16 64 256 1024"
sweep "deferred          " $DEFER    "cifar8w" "16 64 256 1024"
echo "===== C'. other 8-wave shapes, deferred counters (the breakdown to price with)"
sweep "deferred          " $DEFER    "3x3 wino" "1024"
sweep "in-place          " $INPLACE  "3x3 wino" "1024"
