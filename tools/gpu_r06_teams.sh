#!/bin/bash
# round 6: the coarse filter scan as ONE launch of workgroup teams (dph_scan.hip MODE 4) against one launch per 128 rows
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pq.py tests/test_nonfinite.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r06_pytest_pq_teams.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r06_pytest_pq_teams.log
for t in 0 1; do
  DPH_CF_TEAMS=$t timeout 400 python tools/pq_timing.py --nlist 1048576 --batches 64,128,256,512 --steps 10 > gpurun_out/r06_pq_1M_teams$t.log 2>&1; echo "teams=$t exit $?"
  tail -1 gpurun_out/r06_pq_1M_teams$t.log > gpurun_out/r06_pq_ivf1M_170M_timing_teams$t.json
  python - $t <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r06_pq_ivf1M_170M_timing_teams{sys.argv[1]}.json"))
    for b, v in d["batches"].items(): print(f"  B={b}: {v['ms_per_batch']:.3f} ms  {v['queries_per_sec']:.0f} Q/s  coarse {v['coarse_filter_gemm_ms']:.3f} ms  status0 {v['status_zero_rows']} failed_over {v['coarse_failed_over']}")
except Exception as e: print("  parse failed", e)
PY
done
