#!/bin/bash
# round 4, GPU trip 5: what does the staging ds_write cost -- AGPR source? width?  (timing + cycle counters of three more variants)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/scan_diag.py --rows 170000000 --iters 5 --only 0 64 128 68 --out gpurun_out/r04_scan_diag_writes.json 2> gpurun_out/r04_scan_diag_writes.log | tail -c 100
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_scan_diag_writes.json"))
for v in d["variants"]:
    print("  %3d %-46s %s" % (v["bits"], v["variant"], ("%.2f ms  %.3f of int8 peak" % (v["median_ms_after_first"], v["frac_of_5000"])) if "ms" in v else v.get("error","")[-200:]))
PY
for b in 64 128; do
  ( cd /tmp && env DPH_LIBRARY=$R/tools/ubench/libdph_diag$b.so timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/p_v$b -- python $R/tools/scan_diag.py --one --rows 170000000 --n_q 256 --iters 3 > $R/gpurun_out/r04_v$b.log 2>&1 ); echo "variant $b exit $?"
  f=$(find gpurun_out/p_v$b -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/r04_pmc_scan256_variant$b.csv && grep -h "scan_kernel<2" gpurun_out/r04_pmc_scan256_variant$b.csv | cut -c95-200
done
rm -rf gpurun_out/p_*
