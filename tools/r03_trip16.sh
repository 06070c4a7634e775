#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --ignore=tests/test_gpu_search.py --ignore=tests/test_ivf.py --ignore=tests/test_dist_one_gpu.py > gpurun_out/r03_t16_pytest_rest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03_t16_pytest_rest.log
timeout 900 python bench.py > gpurun_out/r03_bench_170M_b64.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_bench_170M_b64.log > gpurun_out/r03_bench_170M_b64.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_bench_170M_b64.json")); r=d["roofline"]
print("Q/s %.1f ms %.3f scan %.3f frac %.4f traffic/alg %s fused %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r.get("traffic_over_algorithmic"), r.get("fused_ladder_stride")))
for k,v in d["also"].items(): print(k, v.get("queries_per_sec"), v.get("ms_per_batch"), v.get("error",""))
print(d["cpu_baseline"]["value"])
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_kt -- python $R/bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_also --no_traffic --recall_queries 0 > $R/gpurun_out/r03_kt_b64.log 2>&1 ); echo "exit $?"
f=$(find gpurun_out/p_kt -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r03_kernel_trace_b64.csv; rm -rf gpurun_out/p_kt; head -5 gpurun_out/r03_kernel_trace_b64.csv | cut -c1-150
timeout 300 python bench.py --no_cpu_baseline --no_also --no_traffic --rows 21250000 --steps 60 --warmup 10 > gpurun_out/r03_bench_21M.log 2>&1; tail -1 gpurun_out/r03_bench_21M.log > gpurun_out/r03_bench_21M_one_of_eight.json
