#!/bin/bash
# round-2 trip 2: where does the time go -- kernel traces of a batch-64 and a batch-128 step, SQ counters of the 256-row scan
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/gpurun_out
echo "== fixed tests"
timeout 300 python -m pytest tests/test_gpu_search.py -m gpu -q --timeout 200 -p no:cacheprovider -k "under_a_bound or mixture" > gpurun_out/t2_pytest.log 2>&1; echo "exit $?"; tail -5 gpurun_out/t2_pytest.log
prof() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 "$@" > $R/gpurun_out/t2_$name.log 2>&1 ); echo "$name exit $?"; }
prof kt64 --kernel-trace --stats -d $R/gpurun_out/t2_kt64 -- python $R/bench.py --steps 6 --warmup 2 --no_cpu_baseline --recall_queries 0
prof kt128 --kernel-trace --stats -d $R/gpurun_out/t2_kt128 -- python $R/bench.py --batch 128 --steps 4 --warmup 2 --no_cpu_baseline --recall_queries 0
prof ktmix --kernel-trace --stats -d $R/gpurun_out/t2_ktmix -- python $R/bench.py --dist mixture --steps 6 --warmup 2 --no_cpu_baseline --recall_queries 0
prof pmcA --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE -d $R/gpurun_out/t2_pmcA -- python $R/bench.py --batch 128 --steps 2 --warmup 1 --no_cpu_baseline --recall_queries 0
prof pmcB --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES -d $R/gpurun_out/t2_pmcB -- python $R/bench.py --batch 128 --steps 2 --warmup 1 --no_cpu_baseline --recall_queries 0
prof pmcA64 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE -d $R/gpurun_out/t2_pmcA64 -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --recall_queries 0
prof fetch128 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/t2_fetch128 -- python $R/bench.py --batch 128 --steps 2 --warmup 1 --no_cpu_baseline --recall_queries 0
cd $R
for d in kt64 kt128 ktmix; do f=$(find gpurun_out/t2_$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/t2_$d.csv; done
for d in pmcA pmcB pmcA64 fetch128; do f=$(find gpurun_out/t2_$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/t2_$d.csv; done
rm -rf gpurun_out/t2_kt64 gpurun_out/t2_kt128 gpurun_out/t2_ktmix gpurun_out/t2_pmcA gpurun_out/t2_pmcB gpurun_out/t2_pmcA64 gpurun_out/t2_fetch128
head -30 gpurun_out/t2_kt64.csv; grep -h scan gpurun_out/t2_pmcA.csv gpurun_out/t2_pmcB.csv | cut -c1-220
