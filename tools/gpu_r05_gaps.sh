#!/bin/bash
# kernel trace of the driver's form, then the idle time between kernels (tools/gap_analysis.py), default and with EDGEHIP_GRAPH=1
OUT=$PWD/gpurun_out/${1:-r05_gaps}; mkdir -p $OUT; export TMPDIR=/tmp
for mode in ${MODES:-plain}; do
  [ $mode = graph ] && export EDGEHIP_GRAPH=1
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $OUT/$mode -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras > $OUT/$mode.json 2> $OUT/$mode.err )
  DB=$(ls $OUT/$mode/*.db $OUT/$mode/*/*.db 2>/dev/null | head -1)
  echo "== $mode"; python -c "
import json; l=open('$OUT/$mode.json').read(); j=json.loads(l[l.rindex('{\"metric\"'):]); print(j['value'], j['ms_per_step'])"
  python tools/gap_analysis.py $DB 20 | tee $OUT/$mode.gaps.txt | head -40
  rm -rf $OUT/$mode
done
