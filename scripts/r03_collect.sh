#!/bin/bash
# After `gpurun -- bash scripts/r03_profiles.sh`: copy the summaries the documentation and bench.py cite from
# gpurun_out/ (scratch) into profiles/ (tracked).   bash scripts/r03_collect.sh
set -e
cd "$(dirname "$0")/.."
P=gpurun_out/prof_r03; Q=gpurun_out/r03_prof; M=gpurun_out/pmc_r03
cp $P/traffic.json profiles/r03_pmc_traffic_1080p_1000frames.json
cp $P/traffic.txt profiles/r03_pmc_traffic.txt
cp $M/instruction_mix.json profiles/r03_pmc_instruction_mix_1080p_1000frames.json
cat $M/a_summary.txt $M/b_summary.txt > profiles/r03_pmc_instruction_mix.txt
cp $P/stats_kernel_stats.csv profiles/r03_kernel_stats_bench_default.csv
cp $P/textured_kernel_stats.csv profiles/r03_kernel_stats_bench_with_textured_leg.csv
cp $P/pipelined_kernel_stats.csv profiles/r03_kernel_stats_bench_chunks4.csv
grep "^{\"metric" $P/bench_under_rocprof.log | tail -1 > profiles/r03_bench_line_under_rocprof.json
grep "^{\"metric" $P/bench_textured_under_rocprof.log | tail -1 > profiles/r03_bench_line_with_textured_leg_under_rocprof.json
for w in 480p 4k dense; do
  cp $Q/${w}_kernel_stats.csv profiles/r03_kernel_stats_bench_${w}.csv
  tail -1 $Q/bench_${w}.json > profiles/r03_bench_${w}_under_rocprof.json
done
python3 - <<'PY'
import json
for f in ["profiles/r03_bench_line_under_rocprof.json", "profiles/r03_bench_line_with_textured_leg_under_rocprof.json"]:
    d = json.load(open(f))
    print(f, d["value"], d["unit"], d["ms_per_step"], "ms/step")
PY
