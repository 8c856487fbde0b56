#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_run5
mkdir -p $OUT
rm -f $OUT/detector_ext.txt
cd $R
timeout -k 10 300 python -u -m pytest tests/test_detector_ext.py -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1
rc=$?; tail -8 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
for lib in "" "build/exp/liborbamd_dd6f718.so"; do
for det in "20,9,1,1" "20,12,0,1" "20,16,1,0"; do
  for scene in sparse textured; do
    echo "== detector $det scene $scene lib ${lib:-current}" | tee -a $OUT/detector_ext.txt
    ORBAMD_LIB=$lib timeout -k 10 120 python scripts/kbench.py --frames 200 --cap 32768 --scene $scene --detector $det 2>&1 | grep -E "fast_mask|mean" | tee -a $OUT/detector_ext.txt
  done
done
done
