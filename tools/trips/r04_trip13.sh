#!/bin/bash
# round 4, GPU trip 13: filter GEMM variant 3 (centroids straight into registers): parity, then timing against variant 2
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_pq.py -m gpu -q -x -p no:cacheprovider -k "65536 or fails_over or kernel_matches" > gpurun_out/r04_pq_parity.log 2>&1; rc=$?; tail -3 gpurun_out/r04_pq_parity.log
if [ $rc != 0 ]; then echo "PARITY FAILED (rc $rc)"; grep -n "Error\|error\|assert" gpurun_out/r04_pq_parity.log | head -8 | cut -c1-300; exit 1; fi
for f in 3 4; do
timeout 240 python tools/pq_timing.py --nlist 1048576 --batches 1,64,256 --steps 20 --tune coarse_filter=$f > gpurun_out/r04_pq_v$f.log 2>&1; echo "exit $?"
tail -1 gpurun_out/r04_pq_v$f.log > gpurun_out/r04_pq_ivf1M_170M_timing_filter$f.json
python - $f <<'PY'
import json,sys
d=json.load(open("gpurun_out/r04_pq_ivf1M_170M_timing_filter%s.json"%sys.argv[1]))
for b,v in d["batches"].items(): print("  filter", sys.argv[1], "batch", b, "%.3f ms %.0f Q/s  gemm %.3f ms  failed_over %s" % (v["ms_per_batch"], v["queries_per_sec"], v["coarse_filter_gemm_ms"] or -1, v["coarse_failed_over"]))
PY
done
