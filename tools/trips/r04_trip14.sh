#!/bin/bash
# round 4, GPU trip 14: the PQ index range-sharded over thread ranks (parity), all PQ tests with GEMM variant 3 as the default
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_pq.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r04_pq_parity.log 2>&1; rc=$?; tail -4 gpurun_out/r04_pq_parity.log
if [ $rc != 0 ]; then echo "PARITY FAILED (rc $rc)"; grep -n "Error\|error\|assert" gpurun_out/r04_pq_parity.log | head -12 | cut -c1-300; exit 1; fi
