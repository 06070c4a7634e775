#!/bin/bash
# Round evidence run on an MI355X: parity tests, smoke, bench lines (batch 64 with the CPU baseline; batches 128 / 256 /
# 512; the mixture dump; one shard of eight), the 8-rank emulation, end-to-end MIPS.search, IVF timings, rocprofv3 kernel traces and
# PMC passes (FETCH_SIZE; SQ counters) of the batch-64 and batch-128 steps.  Everything lands in gpurun_out/r02_*.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-all}
if [ "$T" = all ] || [ "$T" = tests ]; then
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r02_pytest_gpu.log
echo "== smoke"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r02_smoke.log
fi
bench() { name=$1; shift; timeout 500 python bench.py "$@" > gpurun_out/r02_bench_$name.log 2>&1; echo "$name exit $?"; tail -1 gpurun_out/r02_bench_$name.log > gpurun_out/r02_bench_$name.json; python - "$name" <<'PY'
import json,sys
try:
    d=json.load(open(f"gpurun_out/r02_bench_{sys.argv[1]}.json")); r=d["roofline"]
    print("   Q/s %.0f  ms/step %.3f  scan %.3f ms  hbm %.3f  batch-hbm %.3f  mfma %.3f  fast %s  recall %s  cpu %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["per_batch"]["frac"], r["mfma_int8"]["frac"], d["certified_by_first_attempt_last_step"], d.get("recall_at_10"), d.get("cpu_baseline", {}).get("value")))
except Exception as e: print("   parse failed", e)
PY
}
if [ "$T" = all ] || [ "$T" = bench ]; then
echo "== bench"
bench 170M_b64
bench 170M_b128 --batch 128 --no_cpu_baseline
bench 170M_b256 --batch 256 --steps 8 --no_cpu_baseline
bench 170M_b512 --batch 512 --steps 4 --warmup 2 --no_cpu_baseline
bench 170M_b64_mixture --dist mixture --no_cpu_baseline
bench 170M_b256_mixture --dist mixture --batch 256 --steps 8 --no_cpu_baseline
bench 21M_one_of_eight --rows 21250000 --steps 60 --warmup 10 --no_cpu_baseline
echo "== 8-rank strong-scaling emulation"
timeout 300 python tools/scale_emulated.py > gpurun_out/r02_scale_emulated.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r02_scale_emulated.log | cut -c1-400
echo "== end-to-end MIPS.search"
timeout 300 python tools/e2e_mips.py > gpurun_out/r02_e2e.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r02_e2e.log | cut -c1-400
echo "== IVF-4096 / nprobe 256: unit scan vs masked scan vs exact at batch 256; the unit scan at other batch sizes"
timeout 300 python tools/ivf_timing.py > gpurun_out/r02_ivf_timing.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r02_ivf_timing.log > gpurun_out/r02_ivf4096_vs_exact_b256.json; cut -c1-600 gpurun_out/r02_ivf4096_vs_exact_b256.json
: > gpurun_out/r02_ivf4096_units_batches.jsonl
for spec in "512 0" "64 0" "8 0" "1 0" "256 16" "512 64"; do set -- $spec
  timeout 200 python tools/ivf_timing.py --batch $1 --skew $2 --only ivf_units 2>/dev/null | tail -1 >> gpurun_out/r02_ivf4096_units_batches.jsonl
done
cut -c90-330 gpurun_out/r02_ivf4096_units_batches.jsonl
echo "== IVF-4096 build of the 170 M-row dump in HBM (assignment + list builder), then IVF vs exact over the built shard"
timeout 400 python tools/ivf_build_timing.py > gpurun_out/r02_ivf_build.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r02_ivf_build.log > gpurun_out/r02_ivf4096_build_170M.json; cut -c1-700 gpurun_out/r02_ivf4096_build_170M.json
fi
prof() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 "$@" > $R/gpurun_out/r02_$name.log 2>&1 ); echo "$name exit $?"; }
if [ "$T" = all ] || [ "$T" = prof ]; then
echo "== rocprofv3"
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE"
SQB="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES"
prof kt_b64 --kernel-trace --stats -d $R/gpurun_out/p_kt_b64 -- python $R/bench.py --steps 8 --warmup 3 --no_cpu_baseline --recall_queries 0
prof kt_b128 --kernel-trace --stats -d $R/gpurun_out/p_kt_b128 -- python $R/bench.py --batch 128 --steps 6 --warmup 2 --no_cpu_baseline --recall_queries 0
prof kt_21M --kernel-trace --stats -d $R/gpurun_out/p_kt_21M -- python $R/bench.py --rows 21250000 --steps 10 --warmup 3 --no_cpu_baseline --recall_queries 0
prof kt_ivf --kernel-trace --stats -d $R/gpurun_out/p_kt_ivf -- python $R/tools/ivf_timing.py --batch 256 --only ivf_units
prof fetch_b64 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/p_fetch_b64 -- python $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --recall_queries 0
prof fetch_b128 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/p_fetch_b128 -- python $R/bench.py --batch 128 --steps 3 --warmup 1 --no_cpu_baseline --recall_queries 0
prof sqA_b64 --kernel-trace --pmc $SQA -d $R/gpurun_out/p_sqA_b64 -- python $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --recall_queries 0
prof sqB_b64 --kernel-trace --pmc $SQB -d $R/gpurun_out/p_sqB_b64 -- python $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --recall_queries 0
prof sqA_b128 --kernel-trace --pmc $SQA -d $R/gpurun_out/p_sqA_b128 -- python $R/bench.py --batch 128 --steps 3 --warmup 1 --no_cpu_baseline --recall_queries 0
prof sqB_b128 --kernel-trace --pmc $SQB -d $R/gpurun_out/p_sqB_b128 -- python $R/bench.py --batch 128 --steps 3 --warmup 1 --no_cpu_baseline --recall_queries 0
for d in kt_b64 kt_b128 kt_21M; do f=$(find gpurun_out/p_$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r02_kernel_trace_${d#kt_}.csv; done
f=$(find gpurun_out/p_kt_ivf -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r02_kernel_trace_ivf_units_b256.csv
for d in fetch_b64 fetch_b128 sqA_b64 sqB_b64 sqA_b128 sqB_b128; do f=$(find gpurun_out/p_$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/r02_pmc_$d.csv; done
rm -rf gpurun_out/p_*
head -8 gpurun_out/r02_kernel_trace_b64.csv | cut -c1-160
grep -h "scan_kernel<1, 4, false, 0>" gpurun_out/r02_pmc_fetch_b64.csv gpurun_out/r02_pmc_sqA_b64.csv | cut -c90-250
fi
