#!/bin/bash
# Round-3 trip 1: the pair pool (shared chunks instead of per-wave regions) against the parity tests, the headline bench,
# the document-ordered dump, and the full-size IVF build on the mixture dump with in-run recall.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== parity subset"
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_ivf.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "not full_size and not large_synthetic" > gpurun_out/r03_t1_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03_t1_pytest.log
echo "== bench configs[1]"
timeout 300 python bench.py --no_cpu_baseline > gpurun_out/r03_t1_bench_b64.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t1_bench_b64.log | cut -c1-700
echo "== bench docruns b64 / b256"
timeout 300 python bench.py --no_cpu_baseline --dist docruns --per_step --steps 6 > gpurun_out/r03_t1_bench_docruns_b64.log 2>&1; echo "exit $?"; tail -8 gpurun_out/r03_t1_bench_docruns_b64.log | cut -c1-600
timeout 300 python bench.py --no_cpu_baseline --dist docruns --batch 256 --steps 4 --warmup 2 > gpurun_out/r03_t1_bench_docruns_b256.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t1_bench_docruns_b256.log | cut -c1-600
echo "== IVF build, r02 setting (iid, random centroids, random queries)"
timeout 400 python tools/ivf_build_timing.py --kind 0 --centroids random --queries random --cooldown 15 > gpurun_out/r03_t1_ivf_build_iid.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t1_ivf_build_iid.log
echo "== IVF build, mixture + k-means + near queries"
timeout 500 python tools/ivf_build_timing.py --kind 1 --centroids kmeans --queries near > gpurun_out/r03_t1_ivf_build_mixture.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t1_ivf_build_mixture.log
