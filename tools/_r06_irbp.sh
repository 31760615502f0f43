#!/bin/bash
# r06: k_fcn_irb (blocks 2-4) as a persistent grid with the next window prefetched, A/B on one box (experiment build: IVF_FCN_IRB_PERSIST=0 launches one workgroup per tile)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
export IVFRONT_LIB=$R/iv_slam_amd/libivfront_exp.so
for b in 16 1 3; do IVF_B=$b python3 $R/tools/fcn_golden_errors_batched.py > $O/gold_irbp_b$b.txt 2>&1; echo "batch $b: $(tail -1 $O/gold_irbp_b$b.txt)"; done
cd /tmp; export TMPDIR=/tmp
for v in 1 0 1 0; do
  rm -rf $O/prof_irbp_$v; mkdir -p $O/prof_irbp_$v
  IVF_FCN_IRB_PERSIST=$v IVF_B=128 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_irbp_$v -o q -- python3 $R/tools/time_fcn.py > $O/prof_irbp_$v/q.log 2>&1 < /dev/null
  echo "IRB_PERSIST=$v: $(grep 'us/image' $O/prof_irbp_$v/q.log | tail -1)"
  python3 $R/tools/show_stats.py $(ls $O/prof_irbp_$v/*kernel_stats.csv | head -1) 2>/dev/null | grep -i "k_fcn_irb<\|k_fcn_stem" | head -5
  rm -f $O/prof_irbp_$v/*_kernel_trace.csv $O/prof_irbp_$v/*agent_info.csv
done
