#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
pr() { tail -1 "$1" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
echo "== pytest subset (lazy kernel via stride 8)"
DPH_PREPASS_STRIDE=8 timeout 300 python -m pytest tests/test_gpu_search.py -m gpu -q --timeout 300 -p no:cacheprovider -k "scan_lists or search_matches or duplicate or large or two_shards" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -4 gpurun_out/pytest_gpu.log
echo "== bench 40M"
timeout 200 python bench.py --rows 40000000 --no_cpu_baseline --steps 8 --warmup 3 > gpurun_out/b40.log 2>&1; echo "exit $?"; pr gpurun_out/b40.log
echo "== bench 170M"
timeout 300 python bench.py --no_cpu_baseline --steps 8 --warmup 3 > gpurun_out/b170.log 2>&1; echo "exit $?"; pr gpurun_out/b170.log
