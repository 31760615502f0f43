#!/bin/bash
# quick per-kernel durations of one bench configuration (serial: no overlap).  usage: quick_stats.sh [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/quick
mkdir -p $O; rm -f $O/*
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o q -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --serial "$@" > $O/q.log 2>&1 < /dev/null
echo "rc=$?"
grep -h '^{"metric"' $O/q.log | cut -c1-400
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
for f in $O/*kernel_stats.csv; do [ -f "$f" ] && head -24 "$f" | cut -d, -f1-4 | sed 's/(.*)"/"/' | cut -c1-120; done
