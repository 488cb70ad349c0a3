#!/bin/bash
# round 4, call n: do the device's chosen roundings (one scale factor w/q_rho instead of the reference's divide / sqrt / seven divides) add
# departures from the reference's trajectories?  The default command's three free-running parity legs with the default library and with
# the reference-order variant (-DEDGEHIP_TVR_REF_ORDER=1), same box, same frames.
set -u
OUT=$PWD/gpurun_out/r04_n; mkdir -p $OUT
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
for n in default reforder; do
  [ $n = reforder ] && cp tools/experiments/bin/libedgehip_reforder.so rebvo_amd/lib/libedgehip.so
  timeout 600 python bench.py 2>$OUT/bench_$n.err > $OUT/bench_$n.json
  python - $n $OUT/bench_$n.json <<'PY'
import sys, json
n, f = sys.argv[1:]
l = open(f).read(); j = json.loads(l[l.index('{'):])
def legs(d, pre=''):
    for k, v in d.items():
        if k == 'free_running_parity':
            yield pre, v
        elif isinstance(v, dict):
            yield from legs(v, pre + k + '.')
print('[%s] %s frames/s  B.try_velrot %s us' % (n, j['value'], j['kernel_us_per_step'].get('B.try_velrot')))
for pre, v in legs(j):
    print('   %-36s checked %d  outside at last frame %d  departures %s  elsewhere %d  max|dVW| inside %.3g' % (
        pre, v['sequences_checked'], v['sequences_outside_tolerance_at_last_frame'],
        [(d['sequence'], d['first_frame_outside_tolerance'], d['knife_edge_frame']) for d in v['departures']],
        v['departures_elsewhere'], v['max_abs_dVW_while_inside_tolerance']))
PY
done 2>&1 | tee $OUT/summary.txt
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
