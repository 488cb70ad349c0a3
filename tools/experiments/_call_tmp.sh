STAGES="extras ranks8" bash tools/gpu_round.sh r06_end2 2>&1 | cut -c1-400 | tail -30
OUT=gpurun_out/r06_end2
timeout 600 python3 bench.py --tracker-f32 > $OUT/bench_tracker_f32.json 2> $OUT/bench_tracker_f32.err; echo "f32 exit $?"
cp bench_extras.json $OUT/bench_tracker_f32_record.json
head -c 600 $OUT/bench_tracker_f32.json; echo
head -c 600 $OUT/bench_with_extras.json; echo
