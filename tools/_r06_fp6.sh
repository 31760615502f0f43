#!/bin/bash
# r06: FP6 A/B -- goldens through the batched path + per-kernel durations at batch 128, levels 0 / 1 / 2 alternated on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
export IVFRONT_LIB=$R/iv_slam_amd/libivfront_exp.so
LEVELS=${LEVELS:-"0 1"}
for v in $LEVELS; do
  IVF_FCN_FP6=$v python3 $R/tools/fcn_golden_errors_batched.py > $O/gold_fp6_$v.txt 2>&1
  tail -1 $O/gold_fp6_$v.txt
done
for b in 1 2 3; do for v in ${SMALL:-}; do IVF_FCN_FP6=$v IVF_B=$b python3 $R/tools/fcn_golden_errors_batched.py 2>&1 | tail -1; done; done
cd /tmp; export TMPDIR=/tmp
for v in $LEVELS $LEVELS; do
  rm -rf $O/prof_fp6_$v; mkdir -p $O/prof_fp6_$v
  IVF_FCN_FP6=$v IVF_B=128 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp6_$v -o q -- python3 $R/tools/time_fcn.py > $O/prof_fp6_$v/q.log 2>&1 < /dev/null
  echo "level $v: $(grep 'us/image' $O/prof_fp6_$v/q.log | tail -1)"
  python3 $R/tools/show_stats.py $(ls $O/prof_fp6_$v/*kernel_stats.csv $O/prof_fp6_$v/*/*kernel_stats.csv 2>/dev/null | head -1) 2>/dev/null | grep -i "irbd4" | head -4
  rm -f $O/prof_fp6_$v/*_kernel_trace.csv $O/prof_fp6_$v/*agent_info.csv
done
