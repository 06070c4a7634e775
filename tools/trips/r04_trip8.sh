#!/bin/bash
# round 4, GPU trip 8: PQ (tile-major centroid image, two codes in flight in the ADC scans): parity FIRST (stop on failure), then timing + trace
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== PQ parity"
timeout 500 python -m pytest tests/test_pq.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r04_pq_parity.log 2>&1; rc=$?; tail -4 gpurun_out/r04_pq_parity.log
if [ $rc != 0 ]; then echo "PARITY FAILED (rc $rc): stopping"; grep -m5 -n "Error\|error\|fault\|Abort" gpurun_out/r04_pq_parity.log | cut -c1-300; exit 1; fi
echo "== PQ timing"
timeout 240 python tools/pq_timing.py --nlist 1048576 --batches 1,8,64,256 --steps 10 > gpurun_out/r04_pq_1M.log 2>&1; rc=$?; echo "exit $rc"
if [ $rc != 0 ]; then tail -5 gpurun_out/r04_pq_1M.log | cut -c1-300; exit 1; fi
tail -1 gpurun_out/r04_pq_1M.log > gpurun_out/r04_pq_ivf1M_170M_timing.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_pq_ivf1M_170M_timing.json"))
for b,v in d["batches"].items(): print("  batch", b, "%.3f ms %.0f Q/s  gemm %.3f ms  failed_over %s cand/row %.0f" % (v["ms_per_batch"], v["queries_per_sec"], v["coarse_filter_gemm_ms"] or -1, v["coarse_failed_over"], v["coarse_candidates_per_row"] or -1))
PY
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_kt_pq -- python $R/tools/pq_timing.py --nlist 1048576 --batches 64 --steps 6 > $R/gpurun_out/r04_kt_pq.log 2>&1 ); echo "trace exit $?"
f=$(find gpurun_out/p_kt_pq -name "*.db" 2>/dev/null | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r04_kernel_trace_pq_1M_b64.csv
python - <<'PY'
import csv
for r in list(csv.reader(open('gpurun_out/r04_kernel_trace_pq_1M_b64.csv')))[1:]:
    try:
        if len(r) > 3 and any(k in r[0] for k in ('dph_','pq_','fillBuffer')) and float(r[3]) > 8: print("  %-60s calls %4s  avg %9s us" % (r[0][:58], r[1], r[3]))
    except ValueError: pass
PY
rm -rf gpurun_out/p_*
