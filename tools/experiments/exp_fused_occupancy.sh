#!/bin/bash
# One or two workgroups of k_stage_a_fused per CU.  Round 6: w = 256 (compile-time instantiation in an EXPERIMENTS build) is two column
# waves + the scan wave + the fit wave = FOUR waves of 256 registers, so TWO workgroups are resident per CU (8 wave slots at that
# register count, ~55 KB of LDS each) — the one configuration in which this kernel can show what a second resident frame buys; w = 384
# (3 + 2 = five waves) fits one per CU whatever its LDS, like 752 (the control).  Unused dynamic LDS on top (EDGEHIP_FUSED_LDS_PAD) forces one.
# Round 5's note: at w = 368 the kernel's LDS (80.8 KB) lets two workgroups share a CU; padding the
# dynamic LDS request (experiments build) forces one.  Same kernel, same frames, same box: what a second resident workgroup buys.
#   build here:  tools/experiments/exp_fused_occupancy.sh build      on the GPU box:  tools/experiments/exp_fused_occupancy.sh
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$1" = build ]; then
  $ROOT/tools/experiments/build_variant.sh fusedexp stage_a_fused.hip -DEDGEHIP_EXPERIMENTS; exit 0
fi
cd "$GRAFT_REPO_ROOT"
cp rebvo_amd/lib/libedgehip.so /tmp/libedgehip_keep.so
cp tools/experiments/bin/libedgehip_fusedexp.so rebvo_amd/lib/libedgehip.so
for W in ${WIDTHS:-256 384 368}; do
  for PAD in 0 90000 0 90000; do
    echo -n "w=$W lds_pad=$PAD  "
    EDGEHIP_FUSED_LDS_PAD=$PAD EDGEHIP_LEVEL_MODE=3 python tools/prof_stage_a.py 2048 $W 480 2>&1 | grep -E "A.fused" | awk '{print $2, $3}'
  done
done
cp /tmp/libedgehip_keep.so rebvo_amd/lib/libedgehip.so
