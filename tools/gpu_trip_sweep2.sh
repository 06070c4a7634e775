#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_kt21" -- python "$OLDPWD/bench.py" --rows 21250000 --steps 20 --warmup 3 --no_cpu_baseline > "$OLDPWD/gpurun_out/prof_kt21.log" 2>&1 ); echo "exit $?"
