#!/bin/bash
# k_try_velrot: where the time goes.  One library per variant (compile-time switch EDGEHIP_TVR_ABL, wrong results by design):
#   tools/experiments/exp_tvr_ablate.sh build      (here, no GPU needed)
#   tools/experiments/exp_tvr_ablate.sh            (on the GPU box)
# bits: 1 no cross-lane reduction of the 28 sums, 2 no fp64 division / square root, 4 no matched-KeyLine gather,
#       8 no field gather, 16 no residual stream
VARIANTS=${VARIANTS:-"0 1 2 4 8 12 16 31"}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$1" = build ]; then
  mkdir -p $ROOT/tools/experiments/bin
  for A in $VARIANTS; do
    ( cd $ROOT/rebvo_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -I../host/include \
        -DEDGEHIP_TVR_ABL=$A -c stage_b.hip -o /tmp/tvr_abl$A.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/experiments/bin/libedgehip_tvr$A.so \
        $(ls ../lib/obj/*.o | grep -v stage_b.o) /tmp/tvr_abl$A.o ) &
  done
  wait; ls -la $ROOT/tools/experiments/bin/ | grep tvr; exit 0
fi
cd "$GRAFT_REPO_ROOT"
cp rebvo_amd/lib/libedgehip.so /tmp/libedgehip_keep.so
mkdir -p gpurun_out
for A in $VARIANTS; do
  cp tools/experiments/bin/libedgehip_tvr$A.so rebvo_amd/lib/libedgehip.so
  echo -n "ABL=$A  "
  timeout 90 python bench.py --no-extras --cpu-frames 0 --steps 20 --warmup 12 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us_per_step']; print('try_velrot us/step', k['B.try_velrot'], 'evals', d['config']['tryvelrot_evals_per_frame'], 'per eval', round(k['B.try_velrot']/max(1,d['config']['tryvelrot_evals_per_frame']),1))" 2>&1 | tee -a gpurun_out/tvr_ablate.log
done
cp /tmp/libedgehip_keep.so rebvo_amd/lib/libedgehip.so
