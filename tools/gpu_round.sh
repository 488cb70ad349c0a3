#!/bin/bash
# One gpurun call: GPU parity tests, the default bench line, a rocprofv3 kernel trace of the same bench
# command, and two separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950).
# usage: tools/gpu_round.sh <tag> [skip-tests]
# Outputs land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
SKIP_TESTS=${2:-}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH_ARGS=${BENCH_ARGS:-}

if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
  tail -5 "$OUT/pytest_gpu.log"
fi

timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?"; tail -1 "$OUT/smoke.log"

timeout 600 python bench.py $BENCH_ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench exit $?"; tail -c 3000 "$OUT/bench.json"

# the other BASELINE configurations and the ImuMode=2 line (bench lines only; their own CPU legs are short)
if [ -z "${SKIP_CONFIGS:-}" ]; then
  timeout 400 python bench.py --config stage_a --no-extras > "$OUT/bench_stage_a.json" 2> "$OUT/bench_stage_a.err"; echo "stage_a exit $?"
  timeout 400 python bench.py --config tum_undistort --no-extras --cpu-procs 0 > "$OUT/bench_tum_undistort.json" 2> "$OUT/bench_tum.err"; echo "tum exit $?"
  timeout 400 python bench.py --imu --no-extras --cpu-frames 0 > "$OUT/bench_imu.json" 2> "$OUT/bench_imu.err"; echo "imu exit $?"
  tail -c 400 "$OUT/bench_stage_a.json"; tail -c 400 "$OUT/bench_tum_undistort.json"; tail -c 400 "$OUT/bench_imu.json"
fi

# kernel trace of the same command (CPU baseline skipped: it is host work and only lengthens the trace)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- \
    python "$GRAFT_REPO_ROOT/bench.py" $BENCH_ARGS --cpu-frames 0 --no-extras > "$OUT/trace_bench.json" 2> "$OUT/trace.err" )
echo "trace exit $?"
DB=$(ls "$OUT"/trace/*.db "$OUT"/trace/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" "$OUT/kernel_stats.txt" > /dev/null
ls "$OUT/trace" | head

# PMC passes: the driver's form of the command (--steps 20 --warmup 5), counters only (no trace domains besides kernel-trace)
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- \
      python "$GRAFT_REPO_ROOT/bench.py" $BENCH_ARGS --steps 20 --warmup 5 --cpu-frames 0 --no-extras --no-roofline-events \
      > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err" )
  echo "pmc $C exit $?"
  python tools/pmc_summary.py "$OUT/pmc_$C" "$OUT/pmc_$C.txt" "$OUT/pmc.json" > /dev/null
  find "$OUT/pmc_$C" -name '*.csv' -size +5M -delete
done
# stamp the batch size the counters were taken at (bench.py only uses them at the same --nseq)
python - "$OUT/pmc.json" <<'PY'
import json, re, sys
p = sys.argv[1]
try:
    js = json.load(open(p))
    line = open(p.replace("pmc.json", "pmc_FETCH_SIZE.json")).read()
    cfg = json.loads(line[line.index("{"):])["config"]
    js["_nseq"] = cfg["sequences_per_launch"]
    js["_kn"] = cfg.get("keylines_per_frame_timed_mean", cfg["keylines_per_frame"])   # the counters average over this run's frames
    json.dump(js, open(p, "w"), indent=0, sort_keys=True)
except Exception as e:
    print("pmc.json not stamped:", e)
PY
# raw traces are large; keep only the summaries
find "$OUT" -name '*.db' -size +20M -delete
du -sh "$OUT"
