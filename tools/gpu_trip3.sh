#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
pr() { tail -1 "$1" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('certified_rows_last_step'))"; }
echo "== pytest -m gpu (stride 1)"
DPH_PREPASS_STRIDE=1 timeout 400 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 gpurun_out/pytest_gpu.log
echo "== bench 100M"
timeout 200 python bench.py --rows 100000000 --no_cpu_baseline --steps 8 --warmup 3 > gpurun_out/b100.log 2>&1; echo "exit $?"; pr gpurun_out/b100.log
echo "== bench 100M stride 256"
DPH_PREPASS_STRIDE=256 timeout 200 python bench.py --rows 100000000 --no_cpu_baseline --steps 8 --warmup 3 > gpurun_out/b100s256.log 2>&1; echo "exit $?"; pr gpurun_out/b100s256.log
echo "== bench 170M"
timeout 300 python bench.py --no_cpu_baseline --steps 8 --warmup 3 > gpurun_out/b170.log 2>&1; echo "exit $?"; pr gpurun_out/b170.log
