#!/bin/bash
# round 4, GPU trip 3: (a) cycle counters of the scan variants (is the feed's cost stalls or clock?), (b) the PQ coarse filter: parity + timing
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== PQ parity"
timeout 1200 python -m pytest tests/test_pq.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
echo "== PQ timing: filter vs chain"
for f in 1 0; do
timeout 600 python tools/pq_timing.py --nlist 1048576 --batches 64,256 --steps 10 --tune coarse_filter=$f > gpurun_out/r04_pq_1M_filter$f.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r04_pq_1M_filter$f.log > gpurun_out/r04_pq_ivf1M_170M_timing_filter$f.json; cut -c1-600 gpurun_out/r04_pq_ivf1M_170M_timing_filter$f.json
done
echo "== PQ kernel trace (filter on)"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_kt_pq -- python $R/tools/pq_timing.py --nlist 1048576 --batches 64 --steps 6 > $R/gpurun_out/r04_kt_pq.log 2>&1 ); echo "exit $?"
f=$(find gpurun_out/p_kt_pq -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r04_kernel_trace_pq_1M_b64.csv; head -16 gpurun_out/r04_kernel_trace_pq_1M_b64.csv | cut -c1-150
echo "== scan variants: cycles"
for b in 0 4 8 12 63; do
  if [ $b = 0 ]; then L=""; else L="DPH_LIBRARY=$R/tools/ubench/libdph_diag$b.so"; fi
  ( cd /tmp && env $L timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/p_v$b -- python $R/tools/scan_diag.py --one --rows 170000000 --n_q 256 --iters 3 > $R/gpurun_out/r04_v$b.log 2>&1 ); echo "variant $b exit $?"
  f=$(find gpurun_out/p_v$b -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/r04_pmc_scan256_variant$b.csv && grep -h "scan_kernel<2" gpurun_out/r04_pmc_scan256_variant$b.csv | cut -c95-200
done
rm -rf gpurun_out/p_*
