#!/bin/bash
# per-kernel FCN durations at batch $1 (default 8)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-8}
O=$R/gpurun_out/quickfcn$B
mkdir -p $O; rm -f $O/*
cd /tmp; export TMPDIR=/tmp
export IVF_B=$B
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o q -- python3 $R/tools/time_fcn.py > $O/q.log 2>&1 < /dev/null
echo "rc=$?"; tail -1 $O/q.log
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
