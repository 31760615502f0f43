#!/bin/bash
# Round-end evidence in one gpurun call: GPU test suite, the four bench lines (with cpu_baseline), per-shape kernel stats + SQ counters
# (tools/profile_round.sh), HBM traffic (tools/pmc_traffic.sh), batch-1 path stats, tracker timing, the two r04 probes.
# Output under gpurun_out/{evidence,prof,traffic,lat}; copy what is judged into profiles/ (tools/collect_profiles.sh).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
E=$R/gpurun_out/evidence
mkdir -p $E; rm -f $E/*
cd $R
timeout 900 python3 -m pytest tests -m gpu -q 2>&1 | tail -3 > $E/pytest_gpu.txt; cat $E/pytest_gpu.txt
for c in 2 1 3 4; do
  timeout 900 python3 bench.py --config $c > $E/bench_config$c.log 2>&1; echo "bench config $c rc=$?"
  grep -h '^{"metric"' $E/bench_config$c.log | tail -1 > $E/bench_config$c.json
  python3 -c "import json,sys; d=json.load(open('$E/bench_config$c.json')); print('config $c', d['value'], d['ms_per_step'], (d.get('fcn_forward') or {}).get('us_per_image'), d.get('latency_ms_batch1'))" 2>&1 | cut -c1-300
done
timeout 300 python3 tools/time_track.py > $E/track_latency.json 2> $E/track_latency.err; tail -1 $E/track_latency.json | cut -c1-200
make -C tools/probe tile_gather fast_ring > /dev/null 2>&1
timeout 120 tools/probe/tile_gather > $E/probe_tile_gather.txt 2>&1; timeout 120 tools/probe/fast_ring > $E/probe_fast_ring.txt 2>&1
bash tools/profile_round.sh > $E/profile_round.log 2>&1; tail -5 $E/profile_round.log
bash tools/pmc_traffic.sh > $E/pmc_traffic.log 2>&1; tail -3 $E/pmc_traffic.log
bash tools/prof_latency.sh > $E/prof_latency.log 2>&1; tail -3 $E/prof_latency.log
bash tools/prof_fcn.sh final > $E/prof_fcn.txt 2>&1; head -20 $E/prof_fcn.txt
ls -la $E
