#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocprofv3 kernel trace (170M rows)"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_kt170" -- python "$OLDPWD/bench.py" --steps 8 --warmup 3 --no_cpu_baseline > "$OLDPWD/gpurun_out/prof_kt170.log" 2>&1 ); echo "exit $?"
echo "== rocprofv3 pmc FETCH_SIZE (170M rows)"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/prof_pmc_fetch" -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no_cpu_baseline > "$OLDPWD/gpurun_out/prof_pmc_fetch.log" 2>&1 ); echo "exit $?"
