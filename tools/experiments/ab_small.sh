#!/bin/bash
# Small-batch latency A/B on one box: ab_small.sh "<lib name or ->[,ENV=VAL...]" ...   (libs: tools/experiments/bin/libedgehip_<name>.so;
# "-" = the library in the tree).  Prints exp_small_batch.py's lines for 1 and 8 sequences per launch.
cd $GRAFT_REPO_ROOT
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
for spec in "$@"; do
  IFS=',' read -ra parts <<< "$spec"
  n=${parts[0]}
  if [ "$n" != "-" ]; then cp tools/experiments/bin/libedgehip_$n.so rebvo_amd/lib/libedgehip.so; else cp /tmp/keep.so rebvo_amd/lib/libedgehip.so; fi
  echo "[$spec]"
  env "${parts[@]:1}" timeout 300 python tools/experiments/exp_small_batch.py ${SIZES:-1 8} 2>&1 | grep "^n="
done
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
