#!/bin/bash
# kernel timeline (start/end per kernel and queue) of a short bench run.  usage: trace_bench.sh NAME [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
O=$R/gpurun_out/trace_$N
mkdir -p $O; rm -f $O/*
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" > $O/t.log 2>&1 < /dev/null
echo "rc=$?"; grep -h '^{"metric"' $O/t.log | cut -c1-160
ls -la $O | head
