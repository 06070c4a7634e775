#!/bin/bash
# round 4, GPU trip 1: where the 256-row pass loses its time (scan variants + SQ counters), first bench line with the new legs
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== scan variants"
timeout 900 python tools/scan_diag.py --rows 170000000 --iters 6 --out gpurun_out/r04_scan_diag.json 2> gpurun_out/r04_scan_diag.log | tail -c 300
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_scan_diag.json"))
for v in d["variants"]:
    print("  %3d %-40s %s" % (v["bits"], v["variant"], ("%.2f ms  %.3f of int8 peak" % (v["median_ms_after_first"], v["frac_of_5000"])) if "ms" in v else v.get("error","")[-200:]))
PY
echo "== counters of the 256-row launch"
( cd /tmp && rocprofv3 -L > $R/gpurun_out/r04_counters_avail.txt 2>&1 )
grep -c . gpurun_out/r04_counters_avail.txt
prof() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/p_$name -- python $R/tools/scan_diag.py --one --rows 170000000 --n_q 256 --iters 3 > $R/gpurun_out/r04_$name.log 2>&1 ); echo "$name exit $?"; f=$(find gpurun_out/p_$name -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/r04_pmc_${name}.csv; }
prof sqA_scan256 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
prof sqB_scan256 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD
prof sqC_scan256 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
rm -rf gpurun_out/p_*
grep -h "scan_kernel<2" gpurun_out/r04_pmc_sq*_scan256.csv | cut -c95-200
echo "== bench (default run)"
timeout 1200 python bench.py > gpurun_out/r04_bench_170M_b64.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/r04_bench_170M_b64.log > gpurun_out/r04_bench_170M_b64.json
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r04_bench_170M_b64.json")); r=d["roofline"]
    print("   Q/s %.0f  ms/step %.3f  scan %.3f ms  hbm %.3f  per_batch %.3f fixed %.2f ms traffic/alg %s recall %s rows %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["per_batch"]["frac"], d["fixed_ms_per_step"], r.get("traffic_over_algorithmic"), d.get("recall_at_10"), d.get("recall_rows_checked")))
    for k, v in d.get("also", {}).items(): print("   also.%s: %s" % (k, json.dumps({a: b for a, b in v.items() if a not in ("workload", "stats_last_call", "roofline")})[:700]))
except Exception as e: print("   parse failed", e)
PY
tail -5 gpurun_out/r04_bench_170M_b64.log | cut -c1-300
