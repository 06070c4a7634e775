#!/bin/bash
# round 6: the PQ search over skewed indexes (units cut by code count) -- parity, then timing + phase clocks per skew
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
RND=${RND:-r06}
if [ "${1:-all}" != timing ]; then
timeout 900 python -m pytest tests/test_pq.py -m gpu -q -x -p no:cacheprovider --durations=8 > gpurun_out/${RND}_pytest_pq.log 2>&1; echo "pytest exit $?"; tail -12 gpurun_out/${RND}_pytest_pq.log
fi
for sk in none giant zipf lognormal; do
  a=""; [ $sk != none ] && a="--skew $sk"
  timeout 400 python tools/pq_timing.py --nlist 1048576 --batches 64,256 --steps 10 --phases $a > gpurun_out/${RND}_pq_1M_$sk.log 2>&1; echo "$sk exit $?"
  tail -1 gpurun_out/${RND}_pq_1M_$sk.log > gpurun_out/${RND}_pq_ivf1M_skew_${sk}_timing.json
  python - "$sk" <<'PY'
import json, sys, os
try:
    d = json.load(open(f"gpurun_out/{os.environ.get('RND','r06')}_pq_ivf1M_skew_{sys.argv[1]}_timing.json"))
    print("  lists", d["list_sizes"])
    for b, v in d["batches"].items():
        p = v["adc_phases"] or {}
        print(f"  B={b}: {v['ms_per_batch']:.3f} ms  {v['queries_per_sec']:.0f} Q/s  codes {v['codes_scored_per_batch']:.3g}  ns/code/wg {v['ns_per_code_per_workgroup']:.2f}  giant-probing rows {v['rows_probing_the_longest_list']}  status0 {v['status_zero_rows']}  coarse {v['coarse_filter_gemm_ms']}")
        if p: print("     adc: busy mean/max", p["busy_us"], "span", p["kernel_span_us"], "units", p["units"], "codes", p["codes"])
except Exception as e: print("  parse failed", e)
PY
done
