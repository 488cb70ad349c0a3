#!/bin/bash
# round 4, call o: the reference's order of roundings with one shared reciprocal for the seven quotients (the new default) against the
# single scale factor (libedgehip_fastscale.so, -DEDGEHIP_TVR_REF_ORDER=0): tracker parity tests, then both through the default command.
set -u
OUT=$PWD/gpurun_out/r04_o; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_stage_b_gpu.py tests/test_pipeline_gpu.py tests/test_knife_edge_gpu.py tests/test_small_batch_gpu.py tests/test_stereo_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
for n in reforder fastscale; do
  [ $n = fastscale ] && cp tools/experiments/bin/libedgehip_fastscale.so rebvo_amd/lib/libedgehip.so
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
