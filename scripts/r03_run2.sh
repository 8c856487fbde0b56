#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_run2
mkdir -p $OUT
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_batch.py -m gpu -x -q --timeout 300 --timeout-method thread -k "ratio or fetch or frame_file or stager or dropin or extract" > $OUT/pytest.log 2>&1
rc=$?; tail -8 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python scripts/dropin_rate.py > $OUT/dropin_rate.txt 2>&1 || { tail -20 $OUT/dropin_rate.txt; exit 1; }
cat $OUT/dropin_rate.txt
