#!/bin/bash
# k_fcn_irbd4w timing ablations (results wrong): IVF_FCN_WABL masks, FCN us/image at batch 128 with the role-specialised kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
for m in "$@"; do echo -n "wabl $m: "; IVFRONT_LIB=$R/iv_slam_amd/libivfront_exp.so IVF_FCN_ROLES=1 IVF_FCN_WABL=$m FCN_CHUNKS=128 python3 $R/tools/time_fcn_batch.py 2>&1 | grep chunk; done
