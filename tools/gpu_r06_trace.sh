#!/bin/bash
# round 6: per-launch event vs rocprofv3 dispatch durations (tools/trace_out.py), four workloads -> gpurun_out/r06_trace_<leg>.json
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for leg in ${@:-flat anisotropic ivf4096 pq}; do
  timeout 1500 python tools/trace_out.py --leg $leg --out gpurun_out/r06_trace_$leg.json > gpurun_out/r06_trace_$leg.log 2>&1; echo "$leg exit $?"; tail -2 gpurun_out/r06_trace_$leg.log | cut -c1-900
done
