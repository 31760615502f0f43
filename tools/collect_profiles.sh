#!/bin/bash
# copy what tools/round_evidence.sh left under gpurun_out/ into profiles/ under this round's names.  usage: collect_profiles.sh [tag]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
T=${1:-r06}; G=gpurun_out
for c in 1 2 3 4; do [ -s $G/evidence/bench_config$c.json ] && cp $G/evidence/bench_config$c.json profiles/${T}_bench_config$c.json; done
for c in 1 2 3 4; do [ -s $G/prof/c${c}s_kernel_stats.csv ] && cp $G/prof/c${c}s_kernel_stats.csv profiles/${T}_kernel_stats_config${c}_serial.csv; done
[ -s $G/prof/c2p_kernel_stats.csv ] && cp $G/prof/c2p_kernel_stats.csv profiles/${T}_kernel_stats_config2_pipelined.csv
[ -s $G/prof/bench_lines.jsonl ] && cp $G/prof/bench_lines.jsonl profiles/${T}_bench_lines.jsonl
[ -s $G/traffic/r04_pmc_hbm_traffic.json ] && cp $G/traffic/r04_pmc_hbm_traffic.json profiles/${T}_pmc_hbm_traffic.json
[ -s $G/traffic/fetch_calibration.json ] && cp $G/traffic/fetch_calibration.json profiles/${T}_fetch_calibration.json
[ -s $G/lat/lat_kernel_stats.csv ] && cp $G/lat/lat_kernel_stats.csv profiles/${T}_latency_kernel_stats.csv
[ -s $G/evidence/probe_tile_gather.txt ] && cp $G/evidence/probe_tile_gather.txt profiles/${T}_probe_tile_gather.txt
[ -s $G/evidence/probe_fast_ring.txt ] && cp $G/evidence/probe_fast_ring.txt profiles/${T}_probe_fast_ring.txt
[ -s $G/evidence/track_latency.json ] && tail -1 $G/evidence/track_latency.json > profiles/${T}_track_latency.json
[ -s $G/evidence/prof_fcn.txt ] && cp $G/evidence/prof_fcn.txt profiles/${T}_fcn_kernels_batch128.txt
[ -s $G/evidence/pytest_gpu.txt ] && cp $G/evidence/pytest_gpu.txt profiles/${T}_pytest_gpu.txt
[ -s $G/prof/sq1_summary.json ] && python3 tools/sq_to_json.py $G/prof $T
ls -la profiles | grep ${T}_
