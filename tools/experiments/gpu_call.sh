#!/bin/bash
# ONE parameterised gpurun call for tuning work (replaces the 37 one-off tools/experiments/gpu_r04_*.sh / gpu_r05_*.sh of rounds 4-5;
# what each of those measured, and with which arguments of this script it is reproduced, is listed in tools/experiments/CALLS.md).
#
#   gpurun -- 'tools/experiments/gpu_call.sh TAG [--tests "FILES / -k EXPR"] [--ab-libs "NAME ..."] [--ab-env "VAR=A VAR=B ..."]
#                                             [--bench-args "..."] [--rounds N] [--groups "A.fused B.try_velrot ..."] [--small "1 8 64"]'
#
#   --tests     python -m pytest <args> -x -q first (parity before timing)          -> gpurun_out/TAG/pytest.log
#   --ab-libs   same-box A/B of whole libraries tools/experiments/bin/libedgehip_<NAME>.so (tools/experiments/build_variant.sh builds them),
#               alternating, N rounds, through the default bench (driver's form, CPU legs off)   -> gpurun_out/TAG/ab_libs.txt
#   --ab-env    the same with environment settings of ONE library ("EDGEHIP_DUAL_INIT=0 EDGEHIP_DUAL_INIT=1")   -> gpurun_out/TAG/ab_env.txt
#   --small     after an env A/B: each setting at these sequences per launch (200 steps)                         -> gpurun_out/TAG/small.txt
# Every line: frames/s, ms per step, and the HIP-event time per step of the kernel groups named by --groups.
set -u
TAG=$1; shift
TESTS=""; LIBS=""; ENVS=""; BARGS="--steps 20 --warmup 5"; ROUNDS=3; GROUPS_="A.fused B.try_velrot B.try_velrot2 B.build_field C.rotate C.directed_matching"; SMALL=""
while [ $# -gt 0 ]; do
  case "$1" in
    --tests) TESTS=$2; shift 2;; --ab-libs) LIBS=$2; shift 2;; --ab-env) ENVS=$2; shift 2;; --bench-args) BARGS=$2; shift 2;;
    --rounds) ROUNDS=$2; shift 2;; --groups) GROUPS_=$2; shift 2;; --small) SMALL=$2; shift 2;;
    *) echo "unknown argument $1"; exit 2;;
  esac
done
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
line() {   # one bench run -> value, ms per step, the named groups
  timeout 400 python bench.py $BARGS --no-extras --cpu-frames 0 2>/dev/null | GROUPS_="$GROUPS_" python -c "
import sys, json, os
l = sys.stdin.read(); j = json.loads(l[l.index('{'):]); k = json.load(open('bench_extras.json'))['kernel_us_per_step']
print(j['value'], j['ms_per_step'], {g: k.get(g) for g in os.environ['GROUPS_'].split()}, j['config'].get('keylines_per_frame_timed_mean'))"
}
if [ -n "$TESTS" ]; then
  timeout 1200 python -m pytest $TESTS -x -q > "$OUT/pytest.log" 2>&1
  echo "tests exit $?"; grep -v "^REBVO" "$OUT/pytest.log" | tail -3
fi
if [ -n "$LIBS" ]; then
  cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
  for r in $(seq $ROUNDS); do for n in $LIBS; do
    cp tools/experiments/bin/libedgehip_$n.so rebvo_amd/lib/libedgehip.so || { echo "cp of variant $n failed"; df -h . /tmp; ls -la rebvo_amd/lib tools/experiments/bin; cp /tmp/keep.so rebvo_amd/lib/libedgehip.so; exit 3; }
    echo -n "[$n]  "; line
  done; done 2>&1 | tee "$OUT/ab_libs.txt"
  cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
fi
if [ -n "$ENVS" ]; then
  for r in $(seq $ROUNDS); do for e in $ENVS; do
    echo -n "[$e]  "; env $e bash -c "$(declare -f line); BARGS='$BARGS' GROUPS_='$GROUPS_' line"
  done; done 2>&1 | tee "$OUT/ab_env.txt"
  if [ -n "$SMALL" ]; then
    for e in $ENVS; do for n in $SMALL; do
      echo -n "[$e nseq $n] "
      env $e timeout 300 python bench.py --nseq $n --steps 200 --warmup 30 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys, json; l = sys.stdin.read(); j = json.loads(l[l.index('{'):]); print(j['value'], j['ms_per_step'])"
    done; done 2>&1 | tee "$OUT/small.txt"
  fi
fi
