#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/gpurun_out
run() { name=$1; shift; timeout 400 python bench.py --steps 10 --warmup 3 --no_cpu_baseline "$@" > gpurun_out/t4_$name.log 2>&1; echo "$name exit $?"; tail -1 gpurun_out/t4_$name.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.readline()); r=d['roofline']
    print('   Q/s %.0f  ms/step %.2f  scan %.2f ms  hbm %.3f  batch-hbm %.3f  mfma %.3f  pairs %d trig %d fast %s recall %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['per_batch']['frac'], r['mfma_int8']['frac'], d['scan_pairs_last_launch'], d['scan_emit_triggers_last_launch'], d['certified_by_first_attempt_last_step'], d.get('recall_at_10', d.get('recall_error'))))
except Exception as e: print('   parse failed', e)
"; }
run b64_mix --dist mixture
run b128_mix --dist mixture --batch 128
run b64
prof() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 "$@" > $R/gpurun_out/t4_$name.log 2>&1 ); echo "$name exit $?"; }
prof kt64 --kernel-trace --stats -d $R/gpurun_out/t4_kt64 -- python $R/bench.py --steps 6 --warmup 2 --no_cpu_baseline --recall_queries 0
prof pmcA --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE -d $R/gpurun_out/t4_pmcA -- python $R/bench.py --batch 128 --steps 2 --warmup 1 --no_cpu_baseline --recall_queries 0
prof pmcB --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS -d $R/gpurun_out/t4_pmcB -- python $R/bench.py --batch 128 --steps 2 --warmup 1 --no_cpu_baseline --recall_queries 0
prof pmcA64 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE -d $R/gpurun_out/t4_pmcA64 -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --recall_queries 0
cd $R
f=$(find gpurun_out/t4_kt64 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/t4_kt64.csv
python - <<'PY'
import sqlite3,glob
f=glob.glob('gpurun_out/t4_kt64/**/*.db',recursive=True)
if f:
    cur=sqlite3.connect(f[0]).cursor()
    rows=list(cur.execute("select name, duration, start from kernels order by start"))
    # one steady-state step: take the last 40 dispatches
    t0=rows[-60][2]
    out=open('gpurun_out/t4_kt64_timeline.csv','w')
    for n,d,s in rows[-60:]:
        out.write(f"{(s-t0)/1e3:.1f},{d/1e3:.1f},\"{n[:70]}\"\n")
PY
for d in pmcA pmcB pmcA64; do f=$(find gpurun_out/t4_$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f gpurun_out/t4_$d.csv; done
rm -rf gpurun_out/t4_kt64 gpurun_out/t4_pmcA gpurun_out/t4_pmcB gpurun_out/t4_pmcA64
grep -h "scan_kernel<2, 4, false, false>" gpurun_out/t4_pmcA.csv gpurun_out/t4_pmcB.csv | cut -c100-200
