#!/bin/bash
# Trip: new parity tests (device/stream forms, clustered rows, step_exact), end-to-end sync vs stream, config-3 emulation.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_search.py -m gpu -q --timeout 200 -p no:cacheprovider -k "device_and_stream or clustered or resolves_uncertified or reference_layout" > gpurun_out/pytest_new.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_new.log
timeout 300 python tools/e2e_mips.py > gpurun_out/e2e.log 2>&1; echo "e2e exit $?"; tail -1 gpurun_out/e2e.log
timeout 300 python tools/config3_emulated.py > gpurun_out/config3.log 2>&1; echo "config3 exit $?"; tail -2 gpurun_out/config3.log
