#!/bin/bash
# r06: two pyramid levels per launch (k_pyr_down2) again, after the staging loads of both kernels were made to be in flight: A/B on one box, experiment build (IVF_PYR_FUSE=0/1)
R=${GRAFT_REPO_ROOT:-/root/repo}
export IVFRONT_LIB=$R/iv_slam_amd/libivfront_exp.so
for c in 1 4 2; do
  for v in 1 0 1 0; do
    echo "== config $c IVF_PYR_FUSE=$v"
    IVF_PYR_FUSE=$v bash $R/tools/quick_stats.sh --config $c 2>&1 | grep "pyr_down\|\"value\"" | sed 's/"unit".*//' | cut -c1-160
  done
done
