#!/bin/bash
# per-kernel times of the FCN alone (batch 128) under rocprofv3: tools/prof_fcn.sh <tag> [ENV=VALUE ...]  ->  gpurun_out/fcnprof_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
for kv in "$@"; do export "$kv"; done
export FCN_CHUNKS=128
O=$R/gpurun_out/fcnprof_$tag; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/tools/time_fcn_batch.py > $O/log.txt 2>&1
python3 - $O/p_kernel_stats.csv $tag <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "fcn" in r["Name"]]
calls = max(int(r["Calls"]) for r in rows if "k_fcn_stem" in r["Name"])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== %s: FCN kernels, forward of 128 images = 2 launch sequences of 64 (r06): %.1f us per image in kernels; per-launch figures below are for 64 images" % (sys.argv[2], tot / calls / 64 / 1e3))
for r in rows[:16]:
    print("%-66s x%d %9.1f us  %5.1f%%" % (r["Name"][8:74], int(r["Calls"]) // calls, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
P
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
