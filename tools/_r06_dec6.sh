#!/bin/bash
# r06: decoder 3x3 with fp6 correction products, A/B on one box: goldens through the batched and the small-batch paths + per-kernel durations
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
export IVFRONT_LIB=$R/iv_slam_amd/libivfront_exp.so
for v in 1 0; do
  for b in 16 1 2 4; do IVF_FCN_DEC6=$v IVF_B=$b python3 $R/tools/fcn_golden_errors_batched.py > $O/gold_dec6_${v}_b$b.txt 2>&1; echo "DEC6=$v batch $b: $(tail -1 $O/gold_dec6_${v}_b$b.txt)"; done
done
cd /tmp; export TMPDIR=/tmp
for v in 1 0 1 0; do
  rm -rf $O/prof_dec6_$v; mkdir -p $O/prof_dec6_$v
  IVF_FCN_DEC6=$v IVF_B=128 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec6_$v -o q -- python3 $R/tools/time_fcn.py > $O/prof_dec6_$v/q.log 2>&1 < /dev/null
  echo "DEC6=$v: $(grep 'us/image' $O/prof_dec6_$v/q.log | tail -1)"
  python3 $R/tools/show_stats.py $(ls $O/prof_dec6_$v/*kernel_stats.csv | head -1) 2>/dev/null | grep -i "conv3x3" | head -3
  rm -f $O/prof_dec6_$v/*_kernel_trace.csv $O/prof_dec6_$v/*agent_info.csv
done
