#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pq.py -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/r03_t14_pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r03_t14_pytest.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_pq -- python $R/tools/pq_timing.py --nlist 1048576 --batches 64 --steps 6 > $R/gpurun_out/r03_t14_pq_prof.log 2>&1 ); echo "exit $?"
f=$(find gpurun_out/p_pq -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r03_kernel_trace_pq_1M_b64.csv
rm -rf gpurun_out/p_pq; head -6 gpurun_out/r03_kernel_trace_pq_1M_b64.csv | cut -c1-150
timeout 600 python tools/pq_timing.py --nlist 1048576 --batches 64,256 > gpurun_out/r03_t14_pq_1M.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t14_pq_1M.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for b,v in d['batches'].items(): print(b, round(v['ms_per_batch'],3), round(v['queries_per_sec']), v['status_zero_rows'])"
