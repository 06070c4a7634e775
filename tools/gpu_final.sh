#!/bin/bash
# Round evidence run on an MI355X: the full parity suite, smoke, the default bench line (also legs, nested FETCH_SIZE pass,
# CPU baselines), bench lines at other batch shapes and dumps, one shard of eight, the 8-rank emulation, the N = 2 rehearsal on one
# GPU, rocprofv3 kernel traces (batch 64 / 256, the PQ leg), SQ counter passes at batch 128, the FETCH_SIZE pass.  Everything lands in
# gpurun_out/${RND}_*; copy what is to be judged into profiles/.  Targets: all | tests | bench | prof | pq | pqphase | pqlds | pqe2e | aniso | extra | final.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-all}
RND=${RND:-r06}          # prefix of everything written under gpurun_out/ (RND=r04 tools/gpu_final.sh ... in the next round)
if [ "$T" = all ] || [ "$T" = tests ]; then
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --durations=25 > gpurun_out/${RND}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${RND}_pytest_gpu.log
echo "== smoke"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${RND}_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/${RND}_smoke.log
fi
bench() { name=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/${RND}_bench_$name.log 2>&1; echo "$name exit $?"; tail -1 gpurun_out/${RND}_bench_$name.log > gpurun_out/${RND}_bench_$name.json; RND=$RND python - "$name" <<'PY'
import json,sys
try:
    import os; d=json.load(open(f"gpurun_out/{os.environ['RND']}_bench_{sys.argv[1]}.json")); r=d["roofline"]
    print("   Q/s %.0f  ms/step %.3f  scan %.3f ms  hbm %.3f  traffic/alg %s  mfma %.3f  fast %s  recall %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r.get("traffic_over_algorithmic"), r["mfma_int8"]["frac"], d["certified_by_first_attempt_last_step"], d.get("recall_at_10")))
    for k, v in d.get("also", {}).items(): print("   also.%s: %s Q/s, %s ms, leg %.1f s %s" % (k, v.get("queries_per_sec"), v.get("ms_per_batch"), v.get("leg_seconds", 0), v.get("error", "")))
    if "cpu_baseline" in d: print("   cpu: %.2f Q/s, %.0f GFLOP/s on %d threads" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["gflops"], d["cpu_baseline"]["cores"]))
except Exception as e: print("   parse failed", e)
PY
}
if [ "$T" = all ] || [ "$T" = bench ]; then
echo "== bench"
bench 170M_b64
bench 170M_b128 --batch 128 --no_cpu_baseline --no_traffic
bench 170M_b256 --batch 256 --steps 8 --no_cpu_baseline --no_traffic
bench 170M_b64_anisotropic --dist anisotropic --no_cpu_baseline --no_also
bench 170M_b256_anisotropic --dist anisotropic --batch 256 --steps 8 --no_cpu_baseline --no_traffic --no_also
bench 170M_b64_k100 --top_k 100 --no_cpu_baseline --no_traffic --no_also
bench 170M_b64_k200 --top_k 200 --no_cpu_baseline --no_traffic --no_also
bench 170M_b64_L20 --max_answer_length 20 --no_cpu_baseline --no_traffic --no_also
bench 170M_b64_mixture --dist mixture --no_cpu_baseline --no_traffic --no_also
bench 170M_b64_docruns --dist docruns --no_cpu_baseline --no_traffic --no_also
bench 170M_b64_fine64 --tune fine_stride=64 --no_cpu_baseline --no_traffic --no_also
bench 21M_one_of_eight --rows 21250000 --steps 60 --warmup 10 --no_cpu_baseline --no_traffic
echo "== N = 2 rehearsal on one GPU (weak: 2 x 162.5 M rows)"
DPH_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 4 --warmup 2 > gpurun_out/${RND}_bench_n2_weak.log 2>&1; echo "exit $?"; tail -1 gpurun_out/${RND}_bench_n2_weak.log > gpurun_out/${RND}_bench_n2_weak_rehearsal_one_gpu.json; cut -c1-300 gpurun_out/${RND}_bench_n2_weak_rehearsal_one_gpu.json
echo "== 8-rank strong-scaling emulation"
timeout 300 python tools/scale_emulated.py > gpurun_out/${RND}_scale_emulated.log 2>&1; echo "exit $?"; tail -1 gpurun_out/${RND}_scale_emulated.log > gpurun_out/${RND}_scale_emulated_8x21M.json; cut -c1-400 gpurun_out/${RND}_scale_emulated_8x21M.json
fi
prof() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 "$@" > $R/gpurun_out/${RND}_$name.log 2>&1 ); echo "$name exit $?"; }
if [ "$T" = final ]; then
# the short list when GPU minutes are scarce: the default bench line, its kernel trace, then the PQ target below
bench 170M_b64
prof kt_b64 --kernel-trace --stats -d $R/gpurun_out/p_kt_b64 -- python $R/bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_also --no_traffic --recall_queries 0
f=$(find gpurun_out/p_kt_b64 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/${RND}_kernel_trace_b64.csv
rm -rf gpurun_out/p_*; head -4 gpurun_out/${RND}_kernel_trace_b64.csv | cut -c1-160
fi
if [ "$T" = pqe2e ] || [ "$T" = final ]; then
echo "== the reference's shipping configuration end to end (MIPS.search over the OPQ96-IVFPQ index) + its kernel trace"
timeout 400 python tools/pq_e2e.py > gpurun_out/${RND}_pq_e2e.log 2>&1; echo "exit $?"; tail -1 gpurun_out/${RND}_pq_e2e.log > gpurun_out/${RND}_pq_e2e_mips_search_170M.json; cut -c1-500 gpurun_out/${RND}_pq_e2e_mips_search_170M.json
prof kt_pq_e2e --kernel-trace --stats -d $R/gpurun_out/p_kt_pq_e2e -- python $R/tools/pq_e2e.py
f=$(find gpurun_out/p_kt_pq_e2e -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/${RND}_kernel_trace_pq_e2e_b64.csv
rm -rf gpurun_out/p_*; grep -h "dph_\|pq_" gpurun_out/${RND}_kernel_trace_pq_e2e_b64.csv | head -20 | cut -c1-90
fi
if [ "$T" = aniso ] || [ "$T" = final ]; then
echo "== the anisotropic (BERT-like) dump: bench line with the FETCH_SIZE pass, kernel trace"
bench 170M_b64_anisotropic --dist anisotropic --no_cpu_baseline --no_also
prof kt_aniso --kernel-trace --stats -d $R/gpurun_out/p_kt_aniso -- python $R/bench.py --dist anisotropic --steps 8 --warmup 3 --no_cpu_baseline --no_also --no_traffic --recall_queries 0
f=$(find gpurun_out/p_kt_aniso -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/${RND}_kernel_trace_b64_anisotropic.csv
rm -rf gpurun_out/p_*; head -5 gpurun_out/${RND}_kernel_trace_b64_anisotropic.csv | cut -c1-160
fi
if [ "$T" = pqphase ]; then
echo "== PQ parity, phase clock of the ADC scan, kernel trace with per-dispatch selection kernels"
timeout 600 python -m pytest tests/test_pq.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 300 python tools/pq_timing.py --nlist 1048576 --batches 64 --steps 10 --phases > gpurun_out/${RND}_pq_phases.log 2>&1; echo "exit $?"; tail -1 gpurun_out/${RND}_pq_phases.log > gpurun_out/${RND}_pq_ivf1M_phases.json; cut -c1-1500 gpurun_out/${RND}_pq_ivf1M_phases.json
prof kt_pq --kernel-trace --stats -d $R/gpurun_out/p_kt_pq -- python $R/tools/pq_timing.py --nlist 1048576 --batches 64 --steps 6
f=$(find gpurun_out/p_kt_pq -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/${RND}_kernel_trace_pq_1M_b64.csv
rm -rf gpurun_out/p_*; python - <<'PY'
import csv, os
for r in csv.reader(open(f"gpurun_out/{os.environ.get('RND', 'r05')}_kernel_trace_pq_1M_b64.csv")):
    if len(r) == 5 and r[1].isdigit() and int(r[1]) in (7, 14): print(f"   {r[0][:56]:56s} n={r[1]:>3s} avg {float(r[3]):8.1f} us")
PY
timeout 300 python tools/pq_timing.py --nlist 4096 --batches 64 --steps 3 > gpurun_out/${RND}_pq_4096.log 2>&1; echo "exit $?"; tail -1 gpurun_out/${RND}_pq_4096.log > gpurun_out/${RND}_pq_ivf4096_170M_timing.json; cut -c1-300 gpurun_out/${RND}_pq_ivf4096_170M_timing.json
fi
if [ "$T" = pqlds ]; then
echo "== LDS counters of the PQ search (a counter pass of its own)"
prof pmc_lds_pq --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $R/gpurun_out/p_pmc_lds_pq -- python $R/tools/pq_timing.py --nlist 1048576 --batches 64 --steps 3
f=$(find gpurun_out/p_pmc_lds_pq -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/${RND}_pmc_lds_pq_1M_b64.csv
rm -rf gpurun_out/p_*; grep -h "pq_adc_rows\|coarse_select\|coarse_scan" gpurun_out/${RND}_pmc_lds_pq_1M_b64.csv | cut -c1-40,100-200
fi
if [ "$T" = pq ] || [ "$T" = final ]; then
echo "== PQ timing: 2^20 lists and 4096 lists"
timeout 300 python tools/pq_timing.py --nlist 1048576 --batches 1,8,64,256 --steps 10 > gpurun_out/${RND}_pq_1M.log 2>&1; echo "exit $?"; tail -1 gpurun_out/${RND}_pq_1M.log > gpurun_out/${RND}_pq_ivf1M_170M_timing.json; cut -c1-300 gpurun_out/${RND}_pq_ivf1M_170M_timing.json
prof kt_pq --kernel-trace --stats -d $R/gpurun_out/p_kt_pq -- python $R/tools/pq_timing.py --nlist 1048576 --batches 64 --steps 6
f=$(find gpurun_out/p_kt_pq -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/${RND}_kernel_trace_pq_1M_b64.csv
rm -rf gpurun_out/p_*; grep -h "dph_\|pq_" gpurun_out/${RND}_kernel_trace_pq_1M_b64.csv | head -14 | cut -c1-70
timeout 300 python tools/pq_timing.py --nlist 4096 --batches 64 --steps 3 > gpurun_out/${RND}_pq_4096.log 2>&1; echo "exit $?"; tail -1 gpurun_out/${RND}_pq_4096.log > gpurun_out/${RND}_pq_ivf4096_170M_timing.json; cut -c1-300 gpurun_out/${RND}_pq_ivf4096_170M_timing.json
fi
if [ "$T" = extra ]; then
echo "== configs[4] with the encoder in the loop"
timeout 600 python tools/encoder_overlap.py > gpurun_out/${RND}_encoder_overlap.log 2>&1; echo "exit $?"; tail -1 gpurun_out/${RND}_encoder_overlap.log > gpurun_out/${RND}_encoder_overlap_b512.json; cut -c1-400 gpurun_out/${RND}_encoder_overlap_b512.json
fi
if [ "$T" = all ] || [ "$T" = prof ]; then
echo "== rocprofv3"
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE"
prof kt_b64 --kernel-trace --stats -d $R/gpurun_out/p_kt_b64 -- python $R/bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_also --no_traffic --recall_queries 0
prof kt_b256 --kernel-trace --stats -d $R/gpurun_out/p_kt_b256 -- python $R/bench.py --batch 256 --steps 4 --warmup 2 --no_cpu_baseline --no_traffic --recall_queries 0
prof sqA_b128 --kernel-trace --pmc $SQA -d $R/gpurun_out/p_sqA_b128 -- python $R/bench.py --batch 128 --steps 3 --warmup 1 --no_cpu_baseline --no_traffic --recall_queries 0
prof fetch_b64 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/p_fetch_b64 -- python $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_also --no_traffic --recall_queries 0
for d in kt_b64 kt_b256; do f=$(find gpurun_out/p_$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/${RND}_kernel_trace_${d#kt_}.csv; done
for d in sqA_b128 fetch_b64; do f=$(find gpurun_out/p_$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/${RND}_pmc_$d.csv; done
rm -rf gpurun_out/p_*
head -8 gpurun_out/${RND}_kernel_trace_b64.csv | cut -c1-160
grep -h "scan_kernel<1, 4, false, 0" gpurun_out/${RND}_pmc_fetch_b64.csv | cut -c90-250
fi
