#!/bin/bash
# HBM traffic evidence on the GPU box: FETCH_SIZE calibration probe + FETCH_SIZE / WRITE_SIZE passes of bench.py (serial,
# 2 launch sequences of 128 pairs), folded into gpurun_out/traffic/r04_pmc_hbm_traffic.json by tools/pmc_to_json.py.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/traffic
mkdir -p $O; rm -rf $O/*
cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o cal -- $R/tools/probe/fetch_calib > $O/cal.log 2>&1
f=$(ls $O/*cal_counter_collection.csv | head -1); python3 $R/tools/pmc_summary.py $f k_read > $O/fetch_calibration.json; cat $O/fetch_calibration.json; rm -f $f
B="python3 $R/tools/traffic_driver.py"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o pf -- $B > $O/pf.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o pw -- $B > $O/pw.log 2>&1; echo "write rc=$?"
python3 $R/tools/pmc_to_json.py $(ls $O/*pf_counter_collection.csv | head -1) $(ls $O/*pw_counter_collection.csv | head -1) 256 $O/r04_pmc_hbm_traffic.json \
  "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- $B ; second pass --pmc WRITE_SIZE (256 images per front-end launch, 64 per FCN launch)" 64
rm -f $O/*counter_collection.csv $O/*_kernel_trace.csv $O/*agent_info.csv
ls -la $O
