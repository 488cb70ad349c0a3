#!/bin/bash
# round 4, call z: matching in one pass (k_fwd_win + k_rotate<out of place> + k_directed<FUSED>; no k_fwd_apply): bit-identity test and the
# suites that cover the frame path, then A/B against EDGEHIP_FUSE_MATCH=0 (same library), twice, same box.
set -u
OUT=$PWD/gpurun_out/r04_z; mkdir -p $OUT
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "one_pass" 2>&1 | tail -15 | tee $OUT/pytest_one_pass.txt
timeout 1200 python -m pytest tests/test_stage_c_gpu.py tests/test_stage_b_gpu.py tests/test_pipeline_gpu.py tests/test_soak_gpu.py tests/test_small_batch_gpu.py tests/test_host_gpu.py tests/test_dataset_gpu.py tests/test_edge_cases_gpu.py tests/test_knife_edge_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest.txt
ab() {
  echo -n "[$1]  "
  EDGEHIP_FUSE_MATCH=$2 BENCH_FORCE_MOVER=0 timeout 300 python bench.py --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k.get(g) for g in ('C.rotate','C.directed_matching','C.forward_match')})"
}
for r in 1 2; do
  ab three_kernels 0
  ab one_pass 1
done 2>&1 | tee $OUT/ab.txt
