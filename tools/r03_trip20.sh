#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pq.py tests/test_reference_callers.py tests/test_scoring.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r03_t20_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03_t20_pytest.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
