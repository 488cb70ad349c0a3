#!/bin/bash
# One gpurun call of a round.  Stages are picked with STAGES (default: all), e.g.
#   STAGES="tests bench" tools/gpu_round.sh r06_a
#   tests   pytest -m gpu                       -> pytest_gpu.log
#   smoke   __graft_entry__.smoke()             -> smoke.log
#   bench   the driver's command                -> bench.json (the one line), bench_extras.json, bench.err, bench_wall_s
#   extras  python bench.py --extras            -> bench_with_extras.json / bench_extras_full.json
#   trace   rocprofv3 --kernel-trace --stats    -> kernel_stats.txt
#   pmc     FETCH_SIZE / WRITE_SIZE passes      -> pmc.json (stamped with the library's source sha, batch size and KeyLine count)
#   ranks8  BENCH_BACKEND=gloo python bench.py --gpus 8 (NO launcher: bench.py starts its own 8 ranks, which share the one GPU)
#           -> bench_8rank_gloo.json: the dry run of the 8-GPU control flow with real contexts
#   sq      SQ counter passes (issue / wait / LDS) of the same command -> sq.json (same stamp)
# Outputs land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-r06}
STAGES=${STAGES:-"tests smoke bench trace pmc sq"}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH_ARGS=${BENCH_ARGS:-}
has() { [[ " $STAGES " == *" $1 "* ]]; }

if has bench; then
  T0=$(date +%s.%N)
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 $BENCH_ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench exit $?"
  T1=$(date +%s.%N)
  python - "$OUT" "$T0" "$T1" <<'PY'
import json, sys
out, t0, t1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
txt = open(out + "/bench.json").read()
lines = [l for l in txt.splitlines() if l.strip()]
print(f"bench wall {t1 - t0:.1f} s, stdout {len(txt)} bytes in {len(lines)} line(s)")
open(out + "/bench_wall_s", "w").write(f"{t1 - t0:.1f}\n")
js = json.loads(lines[-1])
print(json.dumps(js)[:3800])
PY
  cp bench_extras.json "$OUT/bench_extras.json" 2>/dev/null
fi

if has ranks8; then
  T0=$(date +%s.%N)
  BENCH_BACKEND=gloo BENCH_EXTRAS_FILE="$OUT/bench_8rank_gloo_extras.json" timeout 900 python3 bench.py --gpus 8 --nseq ${RANKS8_NSEQ:-128} --steps 20 --warmup 5 \
      > "$OUT/bench_8rank_gloo.json" 2> "$OUT/bench_8rank_gloo.err"
  echo "ranks8 exit $? wall $(python -c "import time;print(round(time.time()-$T0,1))") s"
  head -c 3000 "$OUT/bench_8rank_gloo.json"; echo
  BENCH_BACKEND=nccl timeout 120 python3 bench.py --gpus 8 --steps 2 --warmup 1 > "$OUT/bench_8rank_rccl_on_1gpu.out" 2> "$OUT/bench_8rank_rccl_on_1gpu.err"
  echo "ranks8 over RCCL on a 1-GPU box: exit $? (must be != 0), stdout bytes $(stat -c %s "$OUT/bench_8rank_rccl_on_1gpu.out")"
  tail -2 "$OUT/bench_8rank_rccl_on_1gpu.err"
fi

if has tests; then
  timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
  tail -5 "$OUT/pytest_gpu.log"
fi

if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
  echo "smoke exit $?"; tail -1 "$OUT/smoke.log"
fi

if has extras; then
  timeout 900 python3 bench.py --extras $BENCH_ARGS > "$OUT/bench_with_extras.json" 2> "$OUT/bench_with_extras.err"
  echo "bench --extras exit $?"
  cp bench_extras.json "$OUT/bench_extras_full.json" 2>/dev/null
fi

PROF_ARGS="--steps 20 --warmup 5 --cpu-frames 0 --no-extras"
if has trace; then
  # kernel trace of the same command (CPU legs skipped: host work that only lengthens the trace)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" $BENCH_ARGS $PROF_ARGS > "$OUT/trace_bench.json" 2> "$OUT/trace.err" )
  echo "trace exit $?"
  DB=$(ls "$OUT"/trace/*.db "$OUT"/trace/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" "$OUT/kernel_stats.txt" > /dev/null
  head -30 "$OUT/kernel_stats.txt"
fi

stamp() {   # $1 = json file, $2 = bench line of one of the counter runs
python - "$1" "$2" <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
p, linefile = sys.argv[1], sys.argv[2]
try:
    js = json.load(open(p))
    line = open(linefile).read()
    cfg = json.loads(line[line.rindex('{"metric"'):])["config"]
    js["_nseq"] = cfg["sequences_per_gpu"]
    js["_kn"] = cfg.get("keylines_per_frame_timed_mean", cfg["keylines_per_frame"])   # the counters average over this run's frames
    js["_src_sha"] = bench.library_source_sha()
    js["_command"] = "python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras --no-roofline-events"
    json.dump(js, open(p, "w"), indent=0, sort_keys=True)
    print("stamped", p, js["_nseq"], js["_kn"], js["_src_sha"])
except Exception as e:
    print("not stamped:", p, e)
PY
}

if has pmc; then
  # counters only (no trace domains besides kernel-trace); FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- \
        python "$GRAFT_REPO_ROOT/bench.py" $BENCH_ARGS $PROF_ARGS --no-roofline-events > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err" )
    echo "pmc $C exit $?"
    python tools/pmc_summary.py "$OUT/pmc_$C" "$OUT/pmc_$C.txt" "$OUT/pmc.json" > /dev/null
    find "$OUT/pmc_$C" -name '*.csv' -size +5M -delete
  done
  stamp "$OUT/pmc.json" "$OUT/pmc_FETCH_SIZE.json"
fi

if has sq; then
  i=0
  for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"; do
    i=$((i+1))
    ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/sq_$i" -o pmc -- \
        python "$GRAFT_REPO_ROOT/bench.py" $BENCH_ARGS $PROF_ARGS --no-roofline-events > "$OUT/sq_$i.json" 2> "$OUT/sq_$i.err" )
    echo "sq group $i exit $?"
    python tools/pmc_summary.py "$OUT/sq_$i" "$OUT/sq_$i.txt" "$OUT/sq.json" > /dev/null
    # kernel durations of the counter pass itself (the clock under counters is not the free-running one)
    python tools/pmc_durations.py "$OUT/sq_$i" "$OUT/sq.json" > /dev/null 2>&1
    find "$OUT/sq_$i" -name '*.csv' -size +5M -delete
  done
  stamp "$OUT/sq.json" "$OUT/sq_1.json"
fi
# raw traces are large; keep only the summaries
find "$OUT" -name '*.db' -size +20M -delete
du -sh "$OUT"
