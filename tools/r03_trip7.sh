#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== PQ tests + builder regression"
timeout 900 python -m pytest tests/test_pq.py tests/test_ivf.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "pq or beyond_2_pow_32 or list_builder" > gpurun_out/r03_t7_pytest.log 2>&1; echo "pytest exit $?"; tail -25 gpurun_out/r03_t7_pytest.log
echo "== bench with also legs"
timeout 600 python bench.py > gpurun_out/r03_t7_bench.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t7_bench.log > gpurun_out/r03_t7_bench.json; python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r03_t7_bench.json"))
    print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("recall_at_1"))
    for k,v in d.get("also",{}).items(): print(k, json.dumps(v)[:900])
except Exception as e:
    print("parse failed", e)
PY
tail -3 gpurun_out/r03_t7_bench.log | cut -c1-300
