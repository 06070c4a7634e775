#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
pr() { tail -1 "$1" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
echo "== pytest -m gpu (prepass stride 1: 1M-row test takes the two-level pre-pass, 400K-row test the single-level one)"
DPH_PREPASS_STRIDE=1 timeout 400 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 gpurun_out/pytest_gpu.log
echo "== bench 40M"
timeout 200 python bench.py --rows 40000000 --no_cpu_baseline --steps 8 --warmup 3 > gpurun_out/b40.log 2>&1; echo "exit $?"; pr gpurun_out/b40.log
echo "== bench 170M"
timeout 300 python bench.py --no_cpu_baseline --steps 8 --warmup 3 > gpurun_out/b170.log 2>&1; echo "exit $?"; pr gpurun_out/b170.log
