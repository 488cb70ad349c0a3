#!/bin/bash
# round 4, call x: rotate_keylines without the gather-record rewrite (search_match reads m_m / n_m themselves; edgehip_ctx::rec_stale):
# the tests that cover rotated slots, then A/B against the previous library, twice, same box.
set -u
OUT=$PWD/gpurun_out/r04_x; mkdir -p $OUT
timeout 900 python -m pytest tests/test_stage_c_gpu.py tests/test_stage_b_gpu.py tests/test_pipeline_gpu.py tests/test_imu_gpu.py tests/test_stereo_gpu.py tests/test_soak_gpu.py tests/test_small_batch_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
ab() {
  echo -n "[$1]  "
  BENCH_FORCE_MOVER=0 timeout 300 python bench.py --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k.get(g) for g in ('C.rotate','C.directed_matching','C.forward_match')})"
}
for r in 1 2; do
  cp tools/experiments/bin/libedgehip_prev.so rebvo_amd/lib/libedgehip.so; ab prev
  cp /tmp/keep.so rebvo_amd/lib/libedgehip.so; ab no_record_rewrite
done 2>&1 | tee $OUT/ab.txt
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
