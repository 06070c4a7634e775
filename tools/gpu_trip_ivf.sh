#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
pr() { tail -1 "$1" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('certified_rows_last_step'))"; }
echo "== pytest ivf (defaults: eager masked kernels)"
timeout 300 python -m pytest tests/test_ivf.py -m gpu -q --timeout 200 -p no:cacheprovider -x > gpurun_out/pytest_ivf.log 2>&1; echo "exit $?"; tail -15 gpurun_out/pytest_ivf.log
echo "== pytest ivf (DPH_PREPASS_STRIDE=1: lazy masked kernels)"
DPH_PREPASS_STRIDE=1 timeout 300 python -m pytest tests/test_ivf.py -m gpu -q --timeout 200 -p no:cacheprovider > gpurun_out/pytest_ivf_s1.log 2>&1; echo "exit $?"; tail -15 gpurun_out/pytest_ivf_s1.log
echo "== pytest -m gpu (everything)"
timeout 500 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
echo "== bench 170M"
timeout 300 python bench.py --no_cpu_baseline --steps 8 --warmup 3 > gpurun_out/b170.log 2>&1; echo "exit $?"; pr gpurun_out/b170.log
