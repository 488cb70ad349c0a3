cd $GRAFT_REPO_ROOT
bash tools/experiments/exp_surface_imu_trace.sh > /dev/null 2>&1
REBVO_GROUP_TIMING=3 rebvo_amd/lib/surface_replay /tmp/cfg_imu /tmp/fi.rgb24 24 8 120 1 0.05 --warmup 40 --threads 8 --group g --stagger > /tmp/o.txt 2> /tmp/e.txt
grep "step 1[01][0-9] " /tmp/e.txt | head -8
grep "us per step" /tmp/o.txt | cut -c1-400
