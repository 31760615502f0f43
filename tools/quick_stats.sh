#!/bin/bash
# quick per-kernel durations of one bench configuration (serial: no overlap).  usage: quick_stats.sh [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/quick
mkdir -p $O; rm -f $O/*
cd /tmp; export TMPDIR=/tmp
export IVF_NO_SIDE_BLUR=1      # as tools/profile_round.sh's *_serial runs: comparable with profiles/*_kernel_stats_config*_serial.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o q -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial "$@" > $O/q.log 2>&1 < /dev/null
echo "rc=$?"
grep -h '^{"metric"' $O/q.log | cut -c1-400
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
python3 - "$O/q_kernel_stats.csv" <<'E'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:60]:
    if 'at::' in r['Name'] or 'rocclr' in r['Name']: continue
    print('%-64s %5s %9.1f us'%(r['Name'].split('(')[0][:64] if not r['Name'].startswith('(') else r['Name'][22:86], r['Calls'], float(r['AverageNs'])/1e3))
E
