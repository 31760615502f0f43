#!/bin/bash
# per-kernel times of configs[1] (extract + match, one batch in flight) under rocprofv3: tools/prof_frontend.sh <tag> [ENV=VALUE ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/feprof_$tag; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial --no-introspect > $O/log.txt 2>&1
python3 - $O/p_kernel_stats.csv $tag <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith("ivf::") or r["Name"].startswith("void ivf::")]
calls = max(int(r["Calls"]) for r in rows if "k_fast_nms" in r["Name"])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== %s: front-end kernels per 128 pairs: %.1f us" % (sys.argv[2], tot / calls / 1e3))
for r in rows[:14]:
    print("%-60s x%d %8.1f us" % (r["Name"].replace("void ", "")[:60], int(r["Calls"]) // calls, float(r["AverageNs"]) / 1e3))
P
grep -h '^{"metric"' $O/log.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pairs/s', d['value'])"
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
