#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for V in 0 6 8; do
( cd /tmp && DPH_SCAN_VARIANT=$V timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_LDS -d "$OLDPWD/gpurun_out/pmc3_v$V" -- python "$OLDPWD/bench.py" --rows 20000000 --steps 3 --warmup 1 --no_cpu_baseline --no_check > "$OLDPWD/gpurun_out/pmc3_v$V.log" 2>&1 ); echo "v$V exit $?"
done
