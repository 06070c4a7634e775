#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_ivf.py tests/test_dist_one_gpu.py -m gpu -q -x -k 'not beyond_2_pow_32' --timeout 900 -p no:cacheprovider > gpurun_out/r03_t15_pytest.log 2>&1; echo "pytest exit $?"; tail -6 gpurun_out/r03_t15_pytest.log
for f in 1 0; do timeout 300 python bench.py --no_cpu_baseline --no_also --no_traffic --tune ladder_fuse=$f > gpurun_out/r03_t15_bench_fuse$f.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t15_bench_fuse$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('fuse', r.get('fused_ladder_stride'), 'Q/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'scan', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],4), d['certified_by_first_attempt_last_step'], d.get('recall_at_10'))"; done
timeout 300 python bench.py --no_cpu_baseline --no_also --no_traffic --rows 21250000 --steps 60 --warmup 10 > gpurun_out/r03_t15_bench_21M.log 2>&1; tail -1 gpurun_out/r03_t15_bench_21M.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('21M fuse', r.get('fused_ladder_stride'), 'Q/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3))"
