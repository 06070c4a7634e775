#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python tools/ivf_build_timing.py --kind 3 --centroids kmeans --queries near --sweep 4,16,64,256 --sweep_noise 0.5 --steps 3 > gpurun_out/r03_t18_ivf_sweep.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t18_ivf_sweep.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps(d['sweep'])); print(d['ivf']['ms_per_batch'], d['ivf_vs_exact'])"
