#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ivf.py tests/test_pq.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "not beyond_2_pow_32" > gpurun_out/r03_t11_pytest.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/r03_t11_pytest.log
timeout 600 python tools/pq_timing.py --nlist 1048576 --batches 64,256 > gpurun_out/r03_t11_pq_1M.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t11_pq_1M.log | cut -c1-400
