#!/bin/bash
cd $GRAFT_REPO_ROOT
lscpu | grep -i "numa\|socket" | head; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3
echo "== unpinned process"; python tools/experiments/exp_h2d_rate.py 64 2>&1 | grep -v "^REBVO\|^$"
for cpus in 0-31 64-95; do echo "== taskset -c $cpus"; taskset -c $cpus python tools/experiments/exp_h2d_rate.py 64 2>&1 | grep -v "^REBVO\|^$"; done
