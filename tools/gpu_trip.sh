#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench at two sizes, rocprofv3 kernel trace.  Everything is wrapped in
# `timeout` so a hung kernel cannot take the box down with it; logs land in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -6
nproc; free -g | head -2
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -40 gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -5 gpurun_out/smoke.log
echo "== bench small (20M rows)"
timeout 600 python bench.py --rows 20000000 --steps 10 --warmup 3 --no_cpu_baseline > gpurun_out/bench_20m.log 2>&1; echo "exit $?"; tail -3 gpurun_out/bench_20m.log
echo "== bench full (170M rows)"
timeout 900 python bench.py > gpurun_out/bench_full.log 2>&1; echo "exit $?"; tail -3 gpurun_out/bench_full.log
echo "== rocprofv3 kernel trace (20M rows)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_kt" -- python "$OLDPWD/bench.py" --rows 20000000 --steps 10 --warmup 3 --no_cpu_baseline > "$OLDPWD/gpurun_out/prof_kt.log" 2>&1 ); echo "exit $?"
find gpurun_out/prof_kt -name "*stats*" | head; for f in $(find gpurun_out/prof_kt -name "*kernel_stats.csv" | head -1); do head -12 "$f"; done
