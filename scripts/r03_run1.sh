#!/bin/bash
# Round-3 GPU batch 1: matrix-port micro, the GPU tests, the default bench, a two-rank rehearsal on one GPU.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_run1
mkdir -p $OUT
cd $R
true
true
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1
rc=$?; tail -15 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"])
print("dropin", json.dumps(d["dropin_call"], indent=1))
print({k: v["avg_ms"] for k, v in d["rooflines"].items()})
PY
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --frames 300 --steps 5 --warmup 2 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err || { tail -20 $OUT/bench_2ranks_gloo.err; exit 1; }
cat $OUT/bench_2ranks_gloo.json | python -c "import json,sys; d=json.load(sys.stdin); print({k: d[k] for k in ('value','n_gpus','seam_verified','seam_wire','world_size','rank_devices','ms_per_step')})"
