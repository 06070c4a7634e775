#!/bin/bash
# Trip: parity tests that touch the pre-pass ladder / threshold kernel, then the bench at 170 M and 21.25 M rows.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "not golden" > gpurun_out/pytest_new.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_new.log
DPH_PREPASS_STRIDE=2 timeout 300 python -m pytest tests/test_gpu_search.py -m gpu -q --timeout 200 -p no:cacheprovider -k "search_matches or duplicate or large or two_shards" > gpurun_out/pytest_s2.log 2>&1; echo "pytest s2 exit $?"; tail -2 gpurun_out/pytest_s2.log
timeout 200 python bench.py --no_cpu_baseline > gpurun_out/bench_170M.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_170M.log | cut -c1-330
timeout 200 python bench.py --rows 21250000 --no_cpu_baseline > gpurun_out/bench_21M.log 2>&1; echo "bench21 exit $?"; tail -1 gpurun_out/bench_21M.log | cut -c1-330
