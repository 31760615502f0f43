#!/bin/bash
# For each variant library var/<name>.so: the FCN goldens through it (parity first), then per-kernel times of the FCN alone at batch 128
# under rocprofv3 (tools/prof_fcn.sh).  usage (gpurun): tools/variants_prof.sh name1 name2 ...   -> gpurun_out/variants.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/variants.txt; : > $O
for v in "$@"; do
  echo "######## $v" >> $O
  (cd $R && IVFRONT_LIB=$R/var/$v.so timeout 300 python -m pytest tests/test_gpu_fcn.py -x -q -k "matches_reference_goldens or (variants and default)" 2>&1 | tail -2) >> $O
  bash $R/tools/prof_fcn.sh $v IVFRONT_LIB=$R/var/$v.so >> $O 2>&1
done
cat $O
