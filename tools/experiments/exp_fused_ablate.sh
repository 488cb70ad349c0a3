#!/bin/bash
# Fused stage-A kernel: phase ablations, one library per variant (compile-time switch, so that the compiler drops the dead
# code and the register allocation of what is left is the real one).  Build here (no GPU needed):
#   tools/experiments/exp_fused_ablate.sh build
# then on the GPU box:  tools/experiments/exp_fused_ablate.sh [nseq]
# bits: 1 scan, 2 box chain, 4 gate bits to the fit wave, 8 fit wave + mask rows off, 16 RGB loads, 32 gradient gate + sign balance,
#       64 fit wave: list only, 128 fit wave: no KeyLine emission, 256 fit wave: no plane fit
VARIANTS=${VARIANTS:-"0 1 4 8 12 32 63 62"}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$1" = build ]; then
  mkdir -p $ROOT/tools/experiments/bin
  for A in $VARIANTS; do
    ( cd $ROOT/rebvo_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -I../host/include \
        -DEDGEHIP_FUSED_ABL=$A -c stage_a_fused.hip -o /tmp/fused_abl$A.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/experiments/bin/libedgehip_abl$A.so \
        $(ls ../lib/obj/*.o | grep -v stage_a_fused.o) /tmp/fused_abl$A.o ) &
  done
  wait; ls -la $ROOT/tools/experiments/bin/ | grep abl; exit 0
fi
cd "$GRAFT_REPO_ROOT"
B=${1:-1024}
cp rebvo_amd/lib/libedgehip.so /tmp/libedgehip_keep.so
for A in $VARIANTS; do
  cp tools/experiments/bin/libedgehip_abl$A.so rebvo_amd/lib/libedgehip.so
  echo -n "ABL=$A  "
  EDGEHIP_LEVEL_MODE=3 python tools/prof_stage_a.py $B 2>&1 | grep -E "fused" | awk '{print $2, $3}'
done
cp /tmp/libedgehip_keep.so rebvo_amd/lib/libedgehip.so
