#!/bin/bash
# round 4, GPU trip 6: PQ with the persistent two-chunk-deep filter GEMM; the 2 M-row IVF parity / recall test; the default bench run with every new leg
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== parity: PQ, the new IVF test, the schedules"
timeout 1500 python -m pytest tests/test_pq.py tests/test_ivf.py tests/test_gpu_search.py -m gpu -q -x -p no:cacheprovider -k "pq or 2M_document or staggered or search_matches_oracle or duplicate" 2>&1 | tail -6
echo "== PQ timing"
timeout 600 python tools/pq_timing.py --nlist 1048576 --batches 64,256 --steps 10 > gpurun_out/r04_pq_1M.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r04_pq_1M.log > gpurun_out/r04_pq_ivf1M_170M_timing.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_pq_ivf1M_170M_timing.json"))
for b,v in d["batches"].items(): print("  batch", b, "%.3f ms %.0f Q/s  gemm %.3f ms  failed_over %s cand/row %.0f" % (v["ms_per_batch"], v["queries_per_sec"], v["coarse_filter_gemm_ms"] or -1, v["coarse_failed_over"], v["coarse_candidates_per_row"] or -1))
PY
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_kt_pq -- python $R/tools/pq_timing.py --nlist 1048576 --batches 64 --steps 6 > $R/gpurun_out/r04_kt_pq.log 2>&1 ); echo "trace exit $?"
f=$(find gpurun_out/p_kt_pq -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r04_kernel_trace_pq_1M_b64.csv
python - <<'PY'
import csv
for r in list(csv.reader(open('gpurun_out/r04_kernel_trace_pq_1M_b64.csv')))[1:]:
    if len(r) > 3 and any(k in r[0] for k in ('dph_','pq_','fillBuffer')) and float(r[3]) > 8: print("  %-60s calls %4s  avg %9s us" % (r[0][:58], r[1], r[3]))
PY
rm -rf gpurun_out/p_*
echo "== bench (default run)"
timeout 1500 python bench.py > gpurun_out/r04_bench_170M_b64.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/r04_bench_170M_b64.log > gpurun_out/r04_bench_170M_b64.json
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r04_bench_170M_b64.json")); r=d["roofline"]
    print("   Q/s %.0f  ms/step %.3f  scan %.3f ms  hbm %.3f  per_batch %.3f fixed %.2f ms traffic/alg %s recall %s rows %s wall %.0f s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["per_batch"]["frac"], d["fixed_ms_per_step"], r.get("traffic_over_algorithmic"), d.get("recall_at_10"), d.get("recall_rows_checked"), d["bench_wall_seconds"]))
    for k, v in d.get("also", {}).items(): print("   also.%s: %s" % (k, json.dumps({a: b for a, b in v.items() if a not in ("workload", "stats_last_call")})[:1500]))
    print("   cpu:", json.dumps(d.get("cpu_baseline", {}))[:300])
except Exception as e: print("   parse failed", e)
PY
tail -3 gpurun_out/r04_bench_170M_b64.log | cut -c1-300
