#!/bin/bash
# Builds timing-only ablation variants of k_fcn_dwpw (compile-time mask IVF_DWPW_ABL, see ivf_fcn.hip) into
# iv_slam_amd/csrc/build/variants/ (run in the build container), or times them on the GPU box:
#   tools/dwpw_ablate.sh build 0 1 4 ...      tools/dwpw_ablate.sh run 0 1 4 ...
R=${GRAFT_REPO_ROOT:-/root/repo}; V=$R/iv_slam_amd/csrc/build/variants; mode=$1; shift
mkdir -p $V
if [ "$mode" = build ]; then
  cp $R/iv_slam_amd/libivfront.so $V/lib_product.so
  for m in "$@"; do
    touch $R/iv_slam_amd/csrc/ivf_fcn.hip
    make -C $R/iv_slam_amd/csrc -j4 EXTRA=-DIVF_DWPW_ABL=$m 2>&1 | grep -E "error" ; cp $R/iv_slam_amd/libivfront.so $V/lib_abl$m.so; echo "built $m"
  done
  cp $V/lib_product.so $R/iv_slam_amd/libivfront.so; touch $R/iv_slam_amd/csrc/ivf_fcn.hip
else
  cp $R/iv_slam_amd/libivfront.so /tmp/lib_product.so
  for m in "$@"; do
    cp $V/lib_abl$m.so $R/iv_slam_amd/libivfront.so
    printf "mask %-5s " $m; python3 $R/tools/time_dwpw_abl.py 2>/dev/null | tail -1
  done
  cp /tmp/lib_product.so $R/iv_slam_amd/libivfront.so
fi
