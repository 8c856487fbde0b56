#!/bin/bash
# Round-3 experiment 1: describe and matcher as one role-specialised launch; timing, then SQ counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_e1
mkdir -p $OUT
cd $R
timeout -k 10 200 python3 scripts/coexec_fused.py > $OUT/fused_timing.txt 2>&1 || { tail -20 $OUT/fused_timing.txt; exit 1; }
cat $OUT/fused_timing.txt
cd /tmp && export TMPDIR=/tmp
cd $R
for mode in separate fused4 fused2; do
  timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES -d $OUT -o pmc_$mode --output-format csv -- python3 scripts/coexec_fused.py --reps 1 --only $mode > $OUT/pmc_$mode.log 2>&1 || { tail -5 $OUT/pmc_$mode.log; exit 1; }
  f=$(ls $OUT/pmc_${mode}_counter_collection.csv 2>/dev/null || find $OUT -name "pmc_${mode}*counter_collection.csv" | head -1)
  echo "== $mode" | tee -a $OUT/pmc_summary.txt
  python3 scripts/pmc_summary.py $f describe knn2_mfma | tee -a $OUT/pmc_summary.txt
done
