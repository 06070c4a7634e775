#!/bin/bash
# Final evidence run of a round: parity tests (default + two-level pre-pass forced on small shards), smoke, the bench
# line, end-to-end MIPS.search, rocprofv3 kernel trace and the FETCH_SIZE pass.  Short timeouts everywhere.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu (defaults)"
timeout 400 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
echo "== pytest -m gpu (DPH_PREPASS_STRIDE=2: two-level pre-pass on the 1M-row test)"
DPH_PREPASS_STRIDE=2 timeout 400 python -m pytest tests/test_gpu_search.py -m gpu -q --timeout 300 -p no:cacheprovider -k "search_matches or duplicate or large or two_shards" > gpurun_out/pytest_gpu_s2.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu_s2.log
echo "== smoke"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
echo "== bench (default: 170M rows, cpu baseline)"
timeout 400 python bench.py > gpurun_out/bench_full.log 2>&1; echo "exit $?"; tail -1 gpurun_out/bench_full.log | cut -c1-1500
echo "== bench at batch 256 (BASELINE configs[3]/[4] batch size; 4 passes of 128 query rows per step)"
timeout 300 python bench.py --batch 256 --steps 6 --warmup 2 --no_cpu_baseline > gpurun_out/bench_b256.log 2>&1; echo "exit $?"; tail -1 gpurun_out/bench_b256.log | cut -c1-260
echo "== 8-rank strong-scaling emulation"
timeout 300 python tools/scale_emulated.py > gpurun_out/scale_emulated.log 2>&1; echo "exit $?"; tail -1 gpurun_out/scale_emulated.log
echo "== end-to-end MIPS.search"
timeout 300 python tools/e2e_mips.py > gpurun_out/e2e.log 2>&1; echo "exit $?"; tail -1 gpurun_out/e2e.log
echo "== rocprofv3 kernel trace (170M rows)"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_kt170" -- python "$OLDPWD/bench.py" --steps 8 --warmup 3 --no_cpu_baseline > "$OLDPWD/gpurun_out/prof_kt170.log" 2>&1 ); echo "exit $?"
echo "== rocprofv3 pmc FETCH_SIZE (170M rows)"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/prof_pmc_fetch" -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no_cpu_baseline > "$OLDPWD/gpurun_out/prof_pmc_fetch.log" 2>&1 ); echo "exit $?"
