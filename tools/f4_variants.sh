#!/bin/bash
# Experiment driver (gpurun): FCN us/image (batch 128, 5 forwards) for each variant library under var/, interleaved twice so that a
# drifting clock shows; then the per-kernel stats of the first variant named in $STATS.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/f4var; mkdir -p $O; rm -f $O/*
cd /tmp; export TMPDIR=/tmp
export IVF_B=128
for round in 1 2; do
  for v in "$@"; do
    IVFRONT_LIB=$R/var/$v.so timeout 200 python3 $R/tools/time_fcn.py 2>&1 | tail -3 | sed "s/^/[$v r$round] /" | tee -a $O/times.log
  done
done
