#!/bin/bash
# round 6: the full GPU suite (or a subset: pass pytest args) -> gpurun_out/r06_pytest_gpu.log
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
if [ $# -gt 0 ]; then
  timeout 2400 python -m pytest "$@" -m gpu -q --timeout 900 -p no:cacheprovider --durations=15 > gpurun_out/r06_pytest_subset.log 2>&1; echo "pytest exit $?"; tail -40 gpurun_out/r06_pytest_subset.log
else
  timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=25 > gpurun_out/r06_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -45 gpurun_out/r06_pytest_gpu.log
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r06_smoke.log
fi
