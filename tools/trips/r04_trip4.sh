#!/bin/bash
# round 4, GPU trip 4: PQ coarse filter after the clean-up (parity, timing of both GEMM variants, kernel trace); cycle counters of the scan variants
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== PQ parity"
timeout 1200 python -m pytest tests/test_pq.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
echo "== PQ timing"
for f in 1 2; do
timeout 600 python tools/pq_timing.py --nlist 1048576 --batches 64,256 --steps 10 --tune coarse_filter=$f > gpurun_out/r04_pq_1M_filter$f.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r04_pq_1M_filter$f.log > gpurun_out/r04_pq_ivf1M_170M_timing_filter$f.json
python - $f <<'PY'
import json,sys
d=json.load(open("gpurun_out/r04_pq_ivf1M_170M_timing_filter%s.json"%sys.argv[1]))
for b,v in d["batches"].items(): print("  filter", sys.argv[1], "batch", b, "%.3f ms %.0f Q/s  gemm %.3f ms  failed_over %s cand/row %.0f" % (v["ms_per_batch"], v["queries_per_sec"], v["coarse_filter_gemm_ms"] or -1, v["coarse_failed_over"], v["coarse_candidates_per_row"] or -1))
PY
done
echo "== PQ kernel trace (filter 1)"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_kt_pq -- python $R/tools/pq_timing.py --nlist 1048576 --batches 64 --steps 6 > $R/gpurun_out/r04_kt_pq.log 2>&1 ); echo "exit $?"
f=$(find gpurun_out/p_kt_pq -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r04_kernel_trace_pq_1M_b64.csv
python - <<'PY'
import csv
for r in list(csv.reader(open('gpurun_out/r04_kernel_trace_pq_1M_b64.csv')))[1:]:
    if len(r) > 3 and any(k in r[0] for k in ('dph_','pq_','fillBuffer')): print("  %-60s calls %4s  avg %9s us" % (r[0][:58], r[1], r[3]))
PY
echo "== scan variants: cycles"
for b in 0 4 8 12 63; do
  if [ $b = 0 ]; then L="X=1"; else L="DPH_LIBRARY=$R/tools/ubench/libdph_diag$b.so"; fi
  ( cd /tmp && env $L timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/p_v$b -- python $R/tools/scan_diag.py --one --rows 170000000 --n_q 256 --iters 3 > $R/gpurun_out/r04_v$b.log 2>&1 ); echo "variant $b exit $?"
  f=$(find gpurun_out/p_v$b -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/r04_pmc_scan256_variant$b.csv && grep -h "scan_kernel<2" gpurun_out/r04_pmc_scan256_variant$b.csv | cut -c95-200
  tail -1 gpurun_out/r04_v$b.log | cut -c1-200
done
rm -rf gpurun_out/p_*
