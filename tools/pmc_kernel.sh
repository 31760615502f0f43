#!/bin/bash
# SQ / TCC counter passes for ONE kernel on a small driver script (default tools/time_fast.py, kernel k_fast_nms).
# usage (on the GPU box, through gpurun): tools/pmc_kernel.sh [kernel-substring] [python script] [tag]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-k_fast_nms}; S=${2:-$R/tools/time_fast.py}; TAG=${3:-pmc}
O=$R/gpurun_out/$TAG
mkdir -p $O; rm -rf $O/*
cd /tmp; export TMPDIR=/tmp
pass() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O -o $n -- python3 $S > $O/$n.log 2>&1 < /dev/null; echo "$n rc=$?"; }
pass sq1 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY
pass sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
pass tf FETCH_SIZE
pass tw WRITE_SIZE
for n in sq1 sq2 tf tw; do f=$(ls $O/*${n}_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $K > $O/$n.json && cat $O/$n.json; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o st -- python3 $S > $O/st.log 2>&1 < /dev/null
f=$(ls $O/*st_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 $f | cut -c1-200
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv $O/*counter_collection.csv
