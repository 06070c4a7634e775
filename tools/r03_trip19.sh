#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/p_pqlds -- python $R/tools/pq_timing.py --nlist 4096 --batches 64 --steps 2 > $R/gpurun_out/r03_t19_pq_lds.log 2>&1 ); echo "exit $?"
f=$(find gpurun_out/p_pqlds -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/r03_pmc_lds_pq_ivf4096.csv; rm -rf gpurun_out/p_pqlds
grep "pq_adc_kernel" gpurun_out/r03_pmc_lds_pq_ivf4096.csv | cut -c1-120
