#!/bin/bash
# A/B helper: build the library of the last commit as iv_slam_amd/libivfront_old.so (select it with IVFRONT_LIB=...), then rebuild the working tree
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
git stash -q; make -C iv_slam_amd/csrc >/dev/null 2>&1; cp iv_slam_amd/libivfront.so iv_slam_amd/libivfront_old.so; git stash pop -q
make -C iv_slam_amd/csrc >/dev/null 2>&1; md5sum iv_slam_amd/libivfront*.so
