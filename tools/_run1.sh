cd $GRAFT_REPO_ROOT
IVFRONT_LIB=$GRAFT_REPO_ROOT/var/stage.so python -m pytest tests/test_gpu_fcn.py -x -q 2>&1 | tail -3
IVFRONT_LIB=$GRAFT_REPO_ROOT/var/late6.so python -m pytest tests/test_gpu_fcn.py -x -q 2>&1 | tail -3
tools/f4_variants.sh pad late5 late6 stage staget stage_nomfma
