#!/bin/bash
# Round-3 trip 3 (the results of trips 1-2 were lost with their container): full parity suite on the pair-pool binary,
# headline bench, document-ordered dump, rocprofv3 kernel trace, full-size IVF build on the mixture dump with k-means
# lists and in-run recall (before / after re-homing the rows).
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r03_pytest_gpu.log
echo "== bench configs[1]"
timeout 400 python bench.py > gpurun_out/r03_bench_170M_b64.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_bench_170M_b64.log | tee gpurun_out/r03_bench_170M_b64.json | cut -c1-900
echo "== bench docruns b64 / b256"
timeout 300 python bench.py --no_cpu_baseline --dist docruns --per_step --steps 6 > gpurun_out/r03_bench_docruns_b64.log 2>&1; echo "exit $?"; tail -8 gpurun_out/r03_bench_docruns_b64.log | cut -c1-600
timeout 300 python bench.py --no_cpu_baseline --dist docruns --batch 256 --steps 4 --warmup 2 > gpurun_out/r03_bench_docruns_b256.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_bench_docruns_b256.log | cut -c1-600
echo "== IVF build: mixture (kind 3) + k-means + near queries, before/after re-home"
timeout 600 python tools/ivf_build_timing.py --kind 3 --centroids kmeans --queries near --before_rehome > gpurun_out/r03_ivf_build_mixture.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_ivf_build_mixture.log | tee gpurun_out/r03_ivf4096_build_170M_mixture.json
echo "== rocprofv3 kernel trace b64"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_kt_b64 -- python $R/bench.py --steps 8 --warmup 3 --no_cpu_baseline --recall_queries 0 > $R/gpurun_out/r03_kt_b64.log 2>&1 ); echo "exit $?"
f=$(find gpurun_out/p_kt_b64 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r03_kernel_trace_b64.csv
find gpurun_out/p_kt_b64 -name "*stats*.csv" | head; rm -rf gpurun_out/p_kt_b64
head -8 gpurun_out/r03_kernel_trace_b64.csv | cut -c1-160
