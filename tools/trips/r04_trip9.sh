#!/bin/bash
# round 4, GPU trip 9: non-temporal loads -- the scan's tile loads (timing variant) and the filter GEMM's centroid stream (coarse_filter=2)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for nq in 128 256; do
timeout 300 python tools/scan_diag.py --rows 170000000 --iters 6 --n_q $nq --only 0 256 --out gpurun_out/r04_scan_diag_nt_$nq.json 2> /dev/null | tail -c 50; echo
python - $nq <<'PY'
import json,sys
d=json.load(open("gpurun_out/r04_scan_diag_nt_%s.json"%sys.argv[1]))
for v in d["variants"]: print("  n_q %s %3d %-28s %s" % (sys.argv[1], v["bits"], v["variant"], ("%.2f ms  hbm %.2f TB/s" % (v["median_ms_after_first"], v["hbm_tb_s"])) if "ms" in v else v.get("error","")[-200:]))
PY
done
for f in 1 2; do
timeout 240 python tools/pq_timing.py --nlist 1048576 --batches 64 --steps 20 --tune coarse_filter=$f > gpurun_out/r04_pq_nt$f.log 2>&1; echo "exit $?"
tail -1 gpurun_out/r04_pq_nt$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for b,v in d['batches'].items(): print('  filter $f batch', b, '%.3f ms %.0f Q/s  gemm %.3f ms' % (v['ms_per_batch'], v['queries_per_sec'], v['coarse_filter_gemm_ms'] or -1))"
done
