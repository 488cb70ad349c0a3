#!/bin/bash
# gpurun_out/<tag> (a full tools/gpu_round.sh run) -> the tracked records under profiles/: the stamped counters the bench line reads
# (pmc_latest.json, sq_latest.json) and the round's final bench line, kernel stats, counter tables, GPU suite tail and smoke log.
#   tools/install_final_profiles.sh r06_final [r06_final]
set -e
SRC=gpurun_out/$1; P=profiles/${2:-r06_final}
cp $SRC/pmc.json profiles/pmc_latest.json
python tools/sq_summary.py $SRC/sq.json profiles/sq_latest.json ${P}_sq_counters.txt > /dev/null
cp $SRC/bench.json ${P}_bench_line.json
cp $SRC/bench_extras.json ${P}_bench_extras.json
cp $SRC/kernel_stats.txt ${P}_kernel_stats_nseq1024.txt
cp $SRC/pmc_FETCH_SIZE.txt ${P}_pmc_FETCH_SIZE.txt
cp $SRC/pmc_WRITE_SIZE.txt ${P}_pmc_WRITE_SIZE.txt
tail -8 $SRC/pytest_gpu.log > ${P}_pytest_gpu_tail.txt
cp $SRC/smoke.log ${P}_smoke.log
python - <<'PY'
import bench
print(bench.pmc_stamp_note())
PY
