#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_run6
mkdir -p $OUT
cd $R
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > $OUT/pytest.log 2>&1
rc=$?; tail -8 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
bash scripts/r03_profiles.sh
