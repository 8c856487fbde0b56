#!/bin/bash
# Collects the profiles that bench.py's roofline block cites: kernel durations of the default
# bench.py run, and HBM traffic per kernel (FETCH_SIZE and WRITE_SIZE in separate PMC passes).
#   bash scripts/profile_round.sh <tag> [round]   (on the GPU box; results under gpurun_out/prof_<tag>/)
set -e
TAG=${1:-run}
ROUND=${2:-r02}     # prefix of the files copied into profiles/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- python3 scripts/kbench.py --frames 1000 --reps 1 > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- python3 scripts/kbench.py --frames 1000 --reps 1 > $OUT/write.log 2>&1
python3 scripts/collect_traffic.py $OUT/fetch_counter_collection.csv $OUT/write_counter_collection.csv $OUT/traffic.json > $OUT/traffic.txt
# bench.py reads the traffic figures from profiles/: use the ones just measured on this build
cp $OUT/traffic.json $R/profiles/${ROUND}_pmc_traffic_1080p_1000frames.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT -o stats --output-format csv -- python3 bench.py --textured-frames 0 --dropin-calls 0 > $OUT/bench_under_rocprof.log 2>&1
# The same with the textured-scene leg (its launches are longer: separate trace so the averages above stay those of the headline workload).
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT -o textured --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --dropin-calls 0 > $OUT/bench_textured_under_rocprof.log 2>&1
# The pipelined variant (frame ranges on two streams), for comparison.
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT -o pipelined --output-format csv -- python3 bench.py --chunks 4 --textured-frames 0 --dropin-calls 0 > $OUT/bench_pipelined_under_rocprof.log 2>&1
cat $OUT/traffic.txt
