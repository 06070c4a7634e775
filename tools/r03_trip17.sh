#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for spec in "170M_b64_mixture --dist mixture" "170M_b256 --batch 256 --steps 8" "170M_b128 --batch 128"; do set -- $spec; name=$1; shift
timeout 400 python bench.py --no_cpu_baseline --no_also --no_traffic "$@" > gpurun_out/r03_bench_$name.log 2>&1; echo "$name exit $?"; tail -1 gpurun_out/r03_bench_$name.log > gpurun_out/r03_bench_$name.json
python - "$name" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r03_bench_{sys.argv[1]}.json")); r=d["roofline"]
print("   Q/s %.0f ms %.3f scan %.3f hbm %.3f mfma %.3f fused %s fast %s recall %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["mfma_int8"]["frac"], r.get("fused_ladder_stride"), d["certified_by_first_attempt_last_step"], d.get("recall_at_10")))
PY
done
echo "== pytest -m gpu (final binary)"
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03_pytest_gpu.log
