#!/bin/bash
# round 6: full GPU suite, the N = 2 rehearsal on one GPU (per-rank roofline + collective waits), FETCH_SIZE of the PQ search at batch 256 (teams)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_r06_tests.sh
echo "== N = 2 rehearsal on one GPU (weak: 2 x 162.5 M rows do not fit one GPU twice -> 2 x 60 M rows)"
DPH_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --rows 120000000 --steps 6 --warmup 2 > gpurun_out/r06_bench_n2.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r06_bench_n2.log > gpurun_out/r06_bench_n2_rehearsal_one_gpu.json
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06_bench_n2_rehearsal_one_gpu.json")); print("  Q/s", d["value"], "per_rank", json.dumps(d.get("per_rank"))[:900])
except Exception as e: print("  parse failed", e)
PY
echo "== FETCH_SIZE of the PQ search at batch 256, teams on / off"
for t in 1 0; do
  ( cd /tmp && DPH_CF_TEAMS=$t timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/p_pmc_pq$t -- python $R/tools/pq_timing.py --nlist 1048576 --batches 256 --steps 3 > $R/gpurun_out/r06_pmc_pq_teams$t.log 2>&1 ); echo "teams=$t exit $?"
  f=$(find gpurun_out/p_pmc_pq$t -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/r06_pmc_fetch_pq_1M_b256_teams$t.csv
  grep -h "coarse_scan" gpurun_out/r06_pmc_fetch_pq_1M_b256_teams$t.csv | cut -c1-300 | head -4
done
rm -rf gpurun_out/p_*
