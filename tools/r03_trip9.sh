#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== PQ timing, the reference's own configuration: 2^20 lists, nprobe 256"
timeout 900 python tools/pq_timing.py --nlist 1048576 --batches 64,256 > gpurun_out/r03_t9_pq_1M.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t9_pq_1M.log | cut -c1-1500
echo "== e2e (sync search through the device chain)"
timeout 300 python tools/e2e_mips.py > gpurun_out/r03_t9_e2e.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t9_e2e.log | cut -c1-600
echo "== golden + reference-caller tests after the search() re-route"
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_reference_callers.py tests/test_pq.py tests/test_model_facade.py tests/test_eval_loop.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "golden or reference or mips or facade or eval or empty" > gpurun_out/r03_t9_pytest.log 2>&1; echo "pytest exit $?"; tail -6 gpurun_out/r03_t9_pytest.log
