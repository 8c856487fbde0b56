#!/bin/bash
# Evidence run: every figure the documentation quotes, collected in one go on the GPU box (see r03_collect.sh).
#   bash scripts/r03_profiles.sh      (results under gpurun_out/r03_prof/ and gpurun_out/prof_r03/, pmc_r03/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_prof
mkdir -p $OUT
cd $R
# instruction mix first: the bench runs below replay it (and the traffic figures) from profiles/, so both files
# have to be this build's before bench.py starts
rm -f gpurun_out/pmc_r03/instruction_mix.json
bash scripts/pmc_insts.sh r03 > $OUT/pmc_insts.log 2>&1 || { tail -20 $OUT/pmc_insts.log; exit 1; }
cp gpurun_out/pmc_r03/instruction_mix.json profiles/r03_pmc_instruction_mix_1080p_1000frames.json
# traffic + kernel stats of the default bench (+ textured leg, + pipelined variant)
bash scripts/profile_round.sh r03 r03 > $OUT/profile_round.log 2>&1 || { tail -20 $OUT/profile_round.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
cd $R
# other workloads: bench line + kernel stats each
for w in "480p --preset 480p" "4k --preset 4k --frames 200" "dense --workload dense-match"; do
  set -- $w; tag=$1; shift
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT -o $tag --output-format csv -- python3 bench.py "$@" --steps 10 --warmup 3 --textured-frames 0 --dropin-calls 0 --cpu-sample 8 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { tail -5 $OUT/bench_$tag.err; exit 1; }
  tail -1 $OUT/bench_$tag.json | python3 -c "import json,sys; d=json.load(sys.stdin); print('$tag', d['value'], d['unit'], d['ms_per_step'])"
done
# one call of the reference's call shape: kernel + memory-copy trace
timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT -o dropin --output-format csv -- python3 scripts/dropin_rate.py --calls 4 > $OUT/dropin_trace.log 2>&1 || { tail -5 $OUT/dropin_trace.log; exit 1; }
ls $OUT
