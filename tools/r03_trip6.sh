#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/ivf_slow_probe.py --assign random > gpurun_out/r03_t6_probe.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t6_probe.log
timeout 600 python tools/ivf_build_timing.py --kind 3 --centroids kmeans --queries near > gpurun_out/r03_t6_ivf_build_mixture.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t6_ivf_build_mixture.log | tee gpurun_out/r03_ivf4096_build_170M_mixture.json
