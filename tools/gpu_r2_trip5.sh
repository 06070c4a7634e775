#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/gpurun_out
prof() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 "$@" > $R/gpurun_out/t5_$name.log 2>&1 ); echo "$name exit $?"; }
prof ktmix --kernel-trace --stats -d $R/gpurun_out/t5_ktmix -- python $R/bench.py --dist mixture --steps 4 --warmup 2 --no_cpu_baseline --recall_queries 0
cd $R
python - <<'PY'
import sqlite3,glob
f=glob.glob('gpurun_out/t5_ktmix/**/*.db',recursive=True)
if f:
    cur=sqlite3.connect(f[0]).cursor()
    rows=list(cur.execute("select name, duration, start from kernels order by start"))
    t0=rows[-70][2]
    out=open('gpurun_out/t5_ktmix_timeline.csv','w')
    for n,d,s in rows[-70:]:
        out.write(f"{(s-t0)/1e3:.1f},{d/1e3:.1f},\"{n[:70]}\"\n")
PY
rm -rf gpurun_out/t5_ktmix
cat gpurun_out/t5_ktmix_timeline.csv | awk -F, '$2>20' | tail -40
