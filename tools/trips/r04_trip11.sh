#!/bin/bash
# round 4, GPU trip 11: why is the document-ordered dump 5 ms per step slower than in round 3?  per-step times under both schedules
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for s in 1 0; do
echo "== docruns, scan_sched=$s"
timeout 300 python bench.py --dist docruns --tune scan_sched=$s --no_cpu_baseline --no_traffic --no_also --recall_queries 0 --steps 8 --warmup 4 --per_step > gpurun_out/r04_docruns_sched$s.log 2>&1; echo "exit $?"
grep "^step" gpurun_out/r04_docruns_sched$s.log | cut -c1-220
tail -1 gpurun_out/r04_docruns_sched$s.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  Q/s %.0f ms/step %.3f scan %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
