#!/bin/bash
# durations of one kernel's dispatches IN LAUNCH ORDER (first N).  usage: trace_order.sh <kernel-substring> <N> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace2
K=$1; N=$2; shift; shift
mkdir -p $O; rm -f $O/*
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --serial "$@" > $O/t.log 2>&1 < /dev/null
echo "rc=$?"
python3 - "$O/t_kernel_trace.csv" "$K" "$N" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print(" ".join("%.0f" % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows[:int(sys.argv[3])]))
PY
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
