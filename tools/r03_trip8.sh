#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== changed tests"
timeout 1200 python -m pytest tests/test_dist_one_gpu.py tests/test_scoring.py tests/test_ivf.py tests/test_gpu_search.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "one_gpu or scoring or logits or mips_class or sent or golden" > gpurun_out/r03_t8_pytest.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/r03_t8_pytest.log
echo "== bench default (traffic + also + cpu)"
( time timeout 900 python bench.py > gpurun_out/r03_t8_bench.log 2>&1 ) 2>&1 | grep real; tail -1 gpurun_out/r03_t8_bench.log > gpurun_out/r03_t8_bench.json; python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r03_t8_bench.json"))
    r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], r["traffic"], r.get("traffic_over_algorithmic"), r["traffic_note"][-120:])
    print({k:(v.get("queries_per_sec"), v.get("leg_seconds"), v.get("error")) for k,v in d["also"].items()}); print(d["cpu_baseline"]["value"], d["cpu_baseline"]["gflops"])
except Exception as e:
    print("parse failed", e)
PY
tail -2 gpurun_out/r03_t8_bench.log | cut -c1-300
echo "== bench N=2 rehearsal on one GPU (strong, 40 M rows)"
DPH_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --rows 40000000 --steps 5 --warmup 2 > gpurun_out/r03_t8_bench_n2_strong.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t8_bench_n2_strong.log | cut -c1-900
echo "== bench N=2 rehearsal on one GPU (default: weak, 162.5 M rows per rank)"
DPH_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 4 --warmup 2 > gpurun_out/r03_t8_bench_n2_weak.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t8_bench_n2_weak.log | cut -c1-900
echo "== PQ timing"
timeout 900 python tools/pq_timing.py > gpurun_out/r03_t8_pq_timing.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t8_pq_timing.log | cut -c1-2000
