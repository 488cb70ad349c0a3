#!/bin/bash
# PMC counters of every kernel of the default bench workload (short run), one rocprofv3 pass per counter group.
# Output: gpurun_out/pmc_all/summary.txt (per kernel: mean counter values per launch)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_all
mkdir -p $OUT
: > $OUT/summary.txt
i=0
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum" \
         "TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g$i -o pmc -- env PYTHONPATH=$GRAFT_REPO_ROOT python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 9 --cpu-frames 0 --no-roofline-events > $OUT/g$i.log 2>&1 )
  echo "group $i exit $?"
  python - "$OUT/g$i" >> $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: print("no csv in", d); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("edgehip::", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v)/len(v)) for c, v in acc[k].items()}, "launches", len(next(iter(acc[k].values()))))
PY
done
find $OUT -name '*.csv' -size +1M -delete
cat $OUT/summary.txt
