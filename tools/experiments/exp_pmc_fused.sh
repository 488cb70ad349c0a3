#!/bin/bash
# PMC counters of the fused stage-A kernel: what the kernels are busy with.  One pass per counter group.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_fused
mkdir -p $OUT
i=0
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g$i -o pmc -- env PYTHONPATH=$GRAFT_REPO_ROOT EDGEHIP_LEVEL_MODE=3 python $GRAFT_REPO_ROOT/tools/prof_stage_a.py 1024 > $OUT/g$i.log 2>&1 )
  echo "group $i exit $?"
  python - "$OUT/g$i" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: print("no csv"); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "fused" in k:
        print(k, {c: round(sum(v)/len(v)) for c, v in acc[k].items()})
PY
done
find $OUT -name '*.csv' -size +3M -delete
