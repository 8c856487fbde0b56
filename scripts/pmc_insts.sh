#!/bin/bash
# Instruction-mix counters per kernel for one pass over the default workload.
#   bash scripts/pmc_insts.sh <tag>     (on the GPU box; results under gpurun_out/pmc_<tag>/)
set -e
TAG=${1:-insts}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SMEM -d $OUT -o a --output-format csv -- python3 scripts/kbench.py --frames 1000 --reps 1 > $OUT/a.log 2>&1
python3 scripts/pmc_summary.py $OUT/a_counter_collection.csv --json $OUT/instruction_mix.json fast_mask orient_describe describe_tile knn2_mfma emit > $OUT/a_summary.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT -o b --output-format csv -- python3 scripts/kbench.py --frames 1000 --reps 1 > $OUT/b.log 2>&1
python3 scripts/pmc_summary.py $OUT/b_counter_collection.csv --json $OUT/instruction_mix.json fast_mask orient_describe describe_tile knn2_mfma emit > $OUT/b_summary.txt
cat $OUT/a_summary.txt $OUT/b_summary.txt
