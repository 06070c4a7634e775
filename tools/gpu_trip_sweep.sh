#!/bin/bash
# Trip: parity tests that touch the pre-pass ladder / threshold kernel, then the 21.25 M-row (one shard of eight) kernel trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_search.py -m gpu -q --timeout 300 -p no:cacheprovider -k "search_matches or duplicate or large or clustered" > gpurun_out/pytest_new.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_new.log
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_kt21" -- python "$OLDPWD/bench.py" --rows 21250000 --steps 20 --warmup 3 --no_cpu_baseline > "$OLDPWD/gpurun_out/prof_kt21.log" 2>&1 ); echo "exit $?"; grep '"metric"' gpurun_out/prof_kt21.log | cut -c1-230
