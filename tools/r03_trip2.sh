#!/bin/bash
# Round-3 trip 2: scan with the query-fragment loads pinned before the loop, the reference's own callers over libdph, k-means
# in HIP, the re-home experiment on the built 170 M-row list-major shard (mixture dump, k-means lists, in-run recall).
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== bench configs[1]"
timeout 300 python bench.py --no_cpu_baseline > gpurun_out/r03_t2_bench_b64.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t2_bench_b64.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'scan', r['avg_launch_ms'], 'frac', r['frac'])"
echo "== new tests"
timeout 900 python -m pytest tests/test_reference_callers.py tests/test_ivf.py tests/test_gpu_search.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "reference or kmeans or synthetic_fill or golden or list_builder or mips_class" > gpurun_out/r03_t2_pytest.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/r03_t2_pytest.log
echo "== IVF build: mixture (kind 3) + k-means + near queries, before/after re-home"
timeout 600 python tools/ivf_build_timing.py --kind 3 --centroids kmeans --queries near --before_rehome > gpurun_out/r03_t2_ivf_build_mixture.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t2_ivf_build_mixture.log
echo "== IVF build: mixture + outliers (kind 1)"
timeout 600 python tools/ivf_build_timing.py --kind 1 --centroids kmeans --queries near > gpurun_out/r03_t2_ivf_build_mixture_outliers.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t2_ivf_build_mixture_outliers.log
