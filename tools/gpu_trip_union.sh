#!/bin/bash
# Trip: two-phase (union bound) sharded search -- parity on two emulated shards, then the 8-rank strong-scaling emulation.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_search.py -m gpu -q --timeout 300 -p no:cacheprovider -k "union_bound or two_shards or merge_records or duplicate or search_matches" > gpurun_out/pytest_new.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_new.log
timeout 300 python tools/scale_emulated.py > gpurun_out/scale_emulated.log 2>&1; echo "scale exit $?"; tail -3 gpurun_out/scale_emulated.log
