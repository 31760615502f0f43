#!/bin/bash
# per-dispatch durations of ONE kernel, grouped by grid size (= by pyramid level / launch shape).  usage: trace_kernel.sh <kernel-substring> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace1
K=$1; shift
mkdir -p $O; rm -f $O/*
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial "$@" > $O/t.log 2>&1 < /dev/null
echo "rc=$?"
python3 - "$O/t_kernel_trace.csv" "$K" <<'PY'
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r['Kernel_Name']:
        g = (r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'), r.get('Workgroup_Size_X', ''))
        agg[g].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for g, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("grid %s  calls %4d  median %8.1f us  min %8.1f  max %8.1f" % (g, len(v), v[len(v) // 2], v[0], v[-1]))
PY
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
