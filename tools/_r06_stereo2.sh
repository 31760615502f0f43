#!/bin/bash
# r06: k_stereo_match2 (two left keypoints per wave) against k_stereo_match, alternated on one box; experiment build (IVF_STEREO_HALF=0/1)
R=${GRAFT_REPO_ROOT:-/root/repo}
export IVFRONT_LIB=$R/iv_slam_amd/libivfront_exp.so
for c in 1 4 2; do
  for v in 1 0 1 0; do
    echo "== config $c IVF_STEREO_HALF=$v: $(IVF_STEREO_HALF=$v bash $R/tools/quick_stats.sh --config $c 2>&1 | grep 'k_stereo_match' | tr -s ' ')"
  done
done
