cd $GRAFT_REPO_ROOT
IVFRONT_LIB=$GRAFT_REPO_ROOT/var/il2.so python -m pytest tests/test_gpu_fcn.py -x -q 2>&1 | tail -3
tools/f4_variants.sh late5 il il2 il2t
