cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ivf.py tests/test_gpu_search.py -m gpu -x -q --timeout 300 -p no:cacheprovider > gpurun_out/t_ivf.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/t_ivf.log
timeout 200 python tools/assign_timing.py > gpurun_out/assign.log 2>&1; echo "assign exit $?"; tail -1 gpurun_out/assign.log
timeout 200 python bench.py --no_cpu_baseline 2>/dev/null | tail -1 | cut -c1-200
