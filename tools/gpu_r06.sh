#!/bin/bash
# Round-6 evidence on an MI355X, by target (gpurun -- 'bash tools/gpu_r06.sh <target> [args]'); everything lands in gpurun_out/r06_*,
# what is to be judged is copied into profiles/.  `RND=r06 bash tools/gpu_final.sh all` is the round-5 set (tests, bench lines, kernel
# traces, counter passes) on the current binary.
#   tests [pytest args]   the GPU suite (or a subset) + smoke
#   bench [name args..]   one bench.py line with a summary
#   skew                  PQ search over skewed indexes: tools/pq_timing.py --skew none|giant|zipf|lognormal with phase clocks
#   teams                 the coarse filter scan as one launch of workgroup teams against one launch per 128 rows + FETCH_SIZE of both
#   trace [legs..]        tools/trace_out.py: events vs rocprofv3 dispatch durations (flat anisotropic ivf4096 pq)
#   n2                    DPH_BENCH_ONE_GPU=1 bench.py --gpus 2 (per-rank roofline, collective waits)
#   cpu                   the torch.mm CPU comparator alone
#   ab                    16 x 16 x 64 against 32 x 32 x 32 on one box (tools/ab_x16.py; needs tools/ubench/libdph_diag_x16off.so built here)
#   diag                  tools/scan_diag.py variants of the 256-row pass (needs the variant libraries built here: --build)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-tests}; shift || true
case "$T" in
tests)
  if [ $# -gt 0 ]; then
    timeout 2400 python -m pytest "$@" -m gpu -q --timeout 900 -p no:cacheprovider --durations=15 > gpurun_out/r06_pytest_subset.log 2>&1; echo "pytest exit $?"; tail -30 gpurun_out/r06_pytest_subset.log
  else
    timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=25 > gpurun_out/r06_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -32 gpurun_out/r06_pytest_gpu.log
    timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r06_smoke.log
  fi ;;
bench)
  name=${1:-170M_b64}; shift || true
  timeout 1500 python bench.py "$@" > gpurun_out/r06_bench_$name.log 2>&1; echo "$name exit $?"; tail -1 gpurun_out/r06_bench_$name.log > gpurun_out/r06_bench_$name.json
  python - "$name" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r06_bench_{sys.argv[1]}.json")); r = d["roofline"]
    print("   Q/s %.0f  ms/step %.3f (median %.3f, min %.3f)  scan %.3f ms  hbm %.3f  mfma %.3f  traffic/alg %s  fast %s  recall %s  wall %.0f s" % (
        d["value"], d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"], r["avg_launch_ms"], r["frac"], r["mfma_int8"]["frac"], r.get("traffic_over_algorithmic"),
        d["certified_by_first_attempt_last_step"], d.get("recall_at_10"), d.get("bench_wall_seconds", 0)))
    for k, v in d.get("also", {}).items():
        print("   also.%s: %s Q/s, %s ms, leg %.1f s %s" % (k, v.get("queries_per_sec"), v.get("ms_per_batch"), v.get("leg_seconds", 0), v.get("error", "")))
        for kk in ("e2e_mips_search", "b512_document_stream", "giant", "b256", "encoder_like_queries"):
            if kk in v: print("        ." + kk + ": " + json.dumps(v[kk])[:600])
    if "cpu_baseline" in d: print("   cpu:", json.dumps({k: v for k, v in d["cpu_baseline"].items() if k not in ("sample", "alt", "per_block")})[:400])
except Exception as e: print("   parse failed", e)
PY
  ;;
skew)
  for sk in none giant zipf lognormal; do
    a=""; [ $sk != none ] && a="--skew $sk"
    timeout 400 python tools/pq_timing.py --nlist 1048576 --batches 64,256 --steps 10 --phases $a > gpurun_out/r06_pq_1M_$sk.log 2>&1; echo "$sk exit $?"
    tail -1 gpurun_out/r06_pq_1M_$sk.log > gpurun_out/r06_pq_ivf1M_skew_${sk}_timing.json
    python - "$sk" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r06_pq_ivf1M_skew_{sys.argv[1]}_timing.json"))
    print("  lists", d["list_sizes"])
    for b, v in d["batches"].items():
        p = v["adc_phases"] or {}
        print(f"  B={b}: {v['ms_per_batch']:.3f} ms  {v['queries_per_sec']:.0f} Q/s  codes {v['codes_scored_per_batch']:.3g}  ns/code/wg {v['ns_per_code_per_workgroup']:.2f}  rows probing the longest list {v['rows_probing_the_longest_list']}  failed over {v['coarse_failed_over']}")
        if p: print("     adc: busy mean/max", p["busy_us"], "span", p["kernel_span_us"])
except Exception as e: print("  parse failed", e)
PY
  done ;;
teams)
  for t in 0 1; do
    DPH_CF_TEAMS=$t timeout 400 python tools/pq_timing.py --nlist 1048576 --batches 64,128,256,512 --steps 10 > gpurun_out/r06_pq_1M_teams$t.log 2>&1; echo "teams=$t exit $?"
    tail -1 gpurun_out/r06_pq_1M_teams$t.log > gpurun_out/r06_pq_ivf1M_170M_timing_teams$t.json
    ( cd /tmp && DPH_CF_TEAMS=$t timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/p_pmc_pq$t -- python $R/tools/pq_timing.py --nlist 1048576 --batches 256 --steps 3 > $R/gpurun_out/r06_pmc_pq_teams$t.log 2>&1 )
    f=$(find gpurun_out/p_pmc_pq$t -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/r06_pmc_fetch_pq_1M_b256_teams$t.csv
    grep -h "coarse_scan" gpurun_out/r06_pmc_fetch_pq_1M_b256_teams$t.csv | cut -c1-300 | head -2
  done
  rm -rf gpurun_out/p_* ;;
trace)
  for leg in ${@:-flat anisotropic ivf4096 pq}; do
    timeout 1500 python tools/trace_out.py --leg $leg --out gpurun_out/r06_trace_$leg.json > gpurun_out/r06_trace_$leg.log 2>&1; echo "$leg exit $?"; tail -1 gpurun_out/r06_trace_$leg.log | cut -c1-900
  done ;;
n2)
  DPH_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --rows 120000000 --steps 6 --warmup 2 > gpurun_out/r06_bench_n2.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r06_bench_n2.log > gpurun_out/r06_bench_n2_rehearsal_one_gpu.json; cut -c1-600 gpurun_out/r06_bench_n2_rehearsal_one_gpu.json ;;
cpu)
  nproc; timeout 300 python -m oracle.cpu_baseline_torch --gib 8 --budget 12 > gpurun_out/r06_cpu_torch.log 2>&1; echo "exit $?"; tail -6 gpurun_out/r06_cpu_torch.log | cut -c1-600 ;;
ab)
  python tools/ab_x16.py > gpurun_out/r06_scan_ab_x16.json 2> gpurun_out/r06_scan_ab_x16.log; grep -o '"mfma": "[0-9x]*", "n_q": [0-9]*, "rep": [0-9], [^]]*], "median_ms_after_first": [0-9.]*' gpurun_out/r06_scan_ab_x16.json | sed 's/"ms": .*"median/median/' ;;
diag)
  timeout 900 python tools/scan_diag.py --rows 170000000 --only 0 512 1024 63 1087 --out gpurun_out/r06_scan_diag_256rows_variants.json > gpurun_out/r06_scan_diag.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r06_scan_diag.log | cut -c1-600 ;;
*) echo "unknown target $T"; exit 2 ;;
esac
