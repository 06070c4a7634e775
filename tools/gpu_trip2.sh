#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/pytest_gpu.log
for S in 64 0 256 16; do
  echo "== bench 170M stride=$S"
  DPH_PREPASS_STRIDE=$S timeout 900 python bench.py --no_cpu_baseline --steps 8 --warmup 3 > gpurun_out/bench_170m_s$S.log 2>&1; echo "exit $?"; tail -1 gpurun_out/bench_170m_s$S.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
echo "== bench 20M"
timeout 600 python bench.py --rows 20000000 --steps 10 --warmup 3 --no_cpu_baseline > gpurun_out/bench_20m.log 2>&1; echo "exit $?"; tail -1 gpurun_out/bench_20m.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
echo "== rocprofv3 kernel trace (170M rows)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_kt170" -- python "$OLDPWD/bench.py" --steps 8 --warmup 3 --no_cpu_baseline > "$OLDPWD/gpurun_out/prof_kt170.log" 2>&1 ); echo "exit $?"
echo "== rocprofv3 pmc FETCH_SIZE (170M rows)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/prof_pmc_fetch" -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no_cpu_baseline > "$OLDPWD/gpurun_out/prof_pmc_fetch.log" 2>&1 ); echo "exit $?"
echo "== rocprofv3 pmc SQ (20M rows)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d "$OLDPWD/gpurun_out/prof_pmc_sq" -- python "$OLDPWD/bench.py" --rows 20000000 --steps 3 --warmup 1 --no_cpu_baseline > "$OLDPWD/gpurun_out/prof_pmc_sq.log" 2>&1 ); echo "exit $?"
ls gpurun_out/prof_*/*/ 2>/dev/null | head -20
