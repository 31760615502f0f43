#!/bin/bash
# decoder (k_fcn_conv3x3_f6) variants: var/<name>.so built by tools/build_variant.sh <name> -D...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for lib in $R/iv_slam_amd/libivfront.so "$@" $R/iv_slam_amd/libivfront.so "$@"; do
  rm -rf $O/abl; mkdir -p $O/abl
  IVFRONT_LIB=$lib IVF_B=128 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl -o q -- python3 $R/tools/time_fcn.py > $O/abl/q.log 2>&1 < /dev/null
  echo "$(basename $lib): $(python3 $R/tools/show_stats.py $(ls $O/abl/*kernel_stats.csv | head -1) 2>/dev/null | grep -i 'conv3x3' | head -1)"
done
