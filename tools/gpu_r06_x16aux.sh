#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_aux.py tests/test_gpu_search.py tests/test_ivf.py tests/test_pipelined.py -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r06_pytest_x16aux.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/r06_pytest_x16aux.log
b() { name=$1; shift; timeout 600 python bench.py "$@" --no_cpu_baseline --no_traffic --no_also > gpurun_out/r06_bench_$name.log 2>&1; echo "$name exit $?"; tail -1 gpurun_out/r06_bench_$name.log > gpurun_out/r06_bench_$name.json
python - $name <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r06_bench_{sys.argv[1]}.json")); r = d["roofline"]
    print("   Q/s %.0f  ms/step %.3f (median %.3f)  scan %.3f ms  hbm %.3f  mfma %.3f  fast %s  recall %s pairs %s" % (d["value"], d["ms_per_step"], d["ms_per_step_median"], r["avg_launch_ms"], r["frac"], r["mfma_int8"]["frac"], d["certified_by_first_attempt_last_step"], d.get("recall_at_10"), d.get("scan_pairs_last_launch")))
except Exception as e: print("   parse failed", e)
PY
}
b 170M_b64_anisotropic --dist anisotropic --steps 10 --warmup 3
b 170M_b256_anisotropic --dist anisotropic --batch 256 --steps 6 --warmup 2
b 170M_b64_anisotropic_encoder --dist anisotropic --queries encoder --steps 10 --warmup 3
b 170M_b64_docruns_encoder --dist docruns --queries encoder --steps 10 --warmup 3
