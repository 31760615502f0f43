#!/bin/bash
# Round-end evidence run on the GPU box (through gpurun): kernel stats + SQ counter passes of bench.py.  Output: gpurun_out/prof/
# (HBM traffic: tools/pmc_traffic.sh)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="$R/bench.py"
run() { name=$1; shift; timeout 600 rocprofv3 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
run stats_c2_pipelined --kernel-trace --stats --output-format csv -d $O -o c2p -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-extras
# the *_serial runs show per-kernel durations WITHOUT sharing: one batch at a time and the blur back on the batch's own stream (by default it
# overlaps the selection chain, which inflates both)
export IVF_NO_SIDE_BLUR=1
run stats_c2_serial    --kernel-trace --stats --output-format csv -d $O -o c2s -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial
run stats_c1_serial    --kernel-trace --stats --output-format csv -d $O -o c1s -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial --config 1
run stats_c3_serial    --kernel-trace --stats --output-format csv -d $O -o c3s -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial --config 3
run stats_c4_serial    --kernel-trace --stats --output-format csv -d $O -o c4s -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial --config 4
unset IVF_NO_SIDE_BLUR
# (FETCH_SIZE / WRITE_SIZE: tools/pmc_traffic.sh, on a strictly sequential driver -- a FETCH_SIZE pass of bench.py's overlapping streams stalled)
export IVF_NO_SIDE_BLUR=1
run pmc_sq1 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace --output-format csv -d $O -o sq1 -- python3 $B --steps 1 --warmup 1 --batches-per-step 4 --no-cpu-baseline --no-extras --serial
run pmc_sq2 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o sq2 -- python3 $B --steps 1 --warmup 1 --batches-per-step 4 --no-cpu-baseline --no-extras --serial
# summarise the counter passes here (the raw CSVs are tens of MB each: gpurun copies back at most 64 MiB) and drop them
for n in sq1 sq2; do
  f=$(ls $O/*${n}_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f "" > $O/${n}_summary.json && rm -f $f
done
grep -h '^{"metric"' $O/stats_c2_pipelined.log $O/stats_c2_serial.log $O/stats_c1_serial.log $O/stats_c3_serial.log $O/stats_c4_serial.log > $O/bench_lines.jsonl
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv      # keep what is summarised; traces are tens of MB
ls -la $O | head -40
