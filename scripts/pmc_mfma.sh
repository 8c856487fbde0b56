#!/bin/bash
# Matrix-pipe counters per kernel for one pass over the default workload.
#   bash scripts/pmc_mfma.sh <tag>     (on the GPU box; results under gpurun_out/pmc_<tag>/)
set -e
TAG=${1:-mfma}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT -o m --output-format csv -- python3 scripts/kbench.py --frames 1000 --reps 1 > $OUT/m.log 2>&1
python3 scripts/pmc_summary.py $OUT/m_counter_collection.csv knn2_mfma > $OUT/m_summary.txt
cat $OUT/m_summary.txt
