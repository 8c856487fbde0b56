#!/bin/bash
# Round-3 experiment 0: extraction and matching on two streams, alone and together, at several residency mixes.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_e0
mkdir -p $OUT
cd $R
for cfg in "0 0" "14000 0" "0 20000" "14000 20000" "42000 0" "42000 20000"; do
  set -- $cfg
  echo "== describe pad $1, knn2 pad $2" | tee -a $OUT/probe.txt
  ORBAMD_PAD_LDS_DESCRIBE=$1 ORBAMD_PAD_LDS_KNN2=$2 timeout -k 10 120 python3 scripts/coexec_probe.py 2>&1 | tee -a $OUT/probe.txt
done
