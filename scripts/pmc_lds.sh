#!/bin/bash
# LDS counters per kernel (bank conflicts, address conflicts, active cycles) for one pass over the
# default workload.
#   bash scripts/pmc_lds.sh <tag>     (on the GPU box; results under gpurun_out/pmc_<tag>/)
set -e
TAG=${1:-lds}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT -o l --output-format csv -- python3 scripts/kbench.py --frames 1000 --reps 1 > $OUT/l.log 2>&1
python3 scripts/pmc_summary.py $OUT/l_counter_collection.csv fast_mask orient_describe describe_tile knn2_mfma emit > $OUT/l_summary.txt
cat $OUT/l_summary.txt
