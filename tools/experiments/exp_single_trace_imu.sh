#!/bin/bash
# one sequence, ImuMode=2: kernel trace of a steady frame
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/single_trace_imu; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace -o single -- python $GRAFT_REPO_ROOT/bench.py --imu --nseq 1 --steps 60 --warmup 12 --no-extras --cpu-frames 0 > $OUT/run.log 2>&1 )
tail -1 $OUT/run.log | cut -c1-200
DB=$(ls $OUT/trace/*.db $OUT/trace/*/*.db 2>/dev/null | head -1)
python tools/rocpd_timeline.py $DB k_rgb_rowscan > $OUT/timeline.txt; cat $OUT/timeline.txt
find $OUT -name '*.db' -delete
