STAGES="ranks8 extras" bash tools/gpu_round.sh r06_final6 2>&1 | cut -c1-300 | tail -8
timeout 600 python3 bench.py --tracker-f32 > gpurun_out/r06_final6/bench_tracker_f32.json 2> gpurun_out/r06_final6/bench_tracker_f32.err; echo "f32 exit $?"
cp bench_extras.json gpurun_out/r06_final6/bench_tracker_f32_record.json
