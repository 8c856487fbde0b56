#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_run4
mkdir -p $OUT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_kat.py tests/test_gpu_dropin.py -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1
rc=$?; tail -30 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --textured-frames 0 --cpu-sample 24 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
tail -1 $OUT/bench.json | python -c "
import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], {k: round(v['avg_ms'],3) for k,v in d['rooflines'].items()}); print(json.dumps(d['dropin_call'])[:1500])"
