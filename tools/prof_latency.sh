#!/bin/bash
# per-kernel durations of the batch-1 (per-call, drop-in) path: tools/time_latency.py under rocprofv3.  Output: gpurun_out/lat/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lat
mkdir -p $O; rm -f $O/*
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o lat -- python3 $R/tools/time_latency.py > $O/lat.log 2>&1 < /dev/null
echo "rc=$?"; tail -3 $O/lat.log
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
python3 $R/tools/show_stats.py $O/lat_kernel_stats.csv
