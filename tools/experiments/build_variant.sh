#!/bin/bash
# A variant of libedgehip.so with ONE source recompiled under extra flags: build_variant.sh <name> <file.hip> <flags...>
# -> tools/experiments/bin/libedgehip_<name>.so (the other objects are the default build's; run `make -C rebvo_amd/csrc` first)
set -e
cd "$(dirname "$0")/../.."
NAME=$1; SRC=$2; shift 2
C=rebvo_amd/csrc; O=rebvo_amd/lib/obj; mkdir -p tools/experiments/bin /tmp/variant_$NAME
EXTRA=""
[ "$SRC" = "stage_imu.hip" ] && EXTRA="-mllvm -unroll-threshold=2000"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$C -Irebvo_amd/host/include -Wall -Wno-unused-result \
   $EXTRA "$@" -c $C/$SRC -o /tmp/variant_$NAME/${SRC%.hip}.o
OBJS=""
for f in api stage_a stage_a_fused stage_b stage_c stage_imu; do
  if [ "$f.hip" = "$SRC" ]; then OBJS="$OBJS /tmp/variant_$NAME/$f.o"; else OBJS="$OBJS $O/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/bin/libedgehip_$NAME.so $OBJS
ls -la tools/experiments/bin/libedgehip_$NAME.so
