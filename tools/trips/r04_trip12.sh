#!/bin/bash
# round 4, GPU trip 12: the in-pass wide re-select: parity (retry-chain tests + goldens), then the document-ordered dump per step
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_search.py -m gpu -q -x -p no:cacheprovider -k "near_ties or beyond_the_sort or duplicate or lost_pairs or search_matches_oracle or golden or clustered or mixture or device_step or union_bound or large_synthetic" > gpurun_out/r04_reselect_parity.log 2>&1; rc=$?; tail -3 gpurun_out/r04_reselect_parity.log
if [ $rc != 0 ]; then echo "PARITY FAILED"; grep -n "Error\|assert" gpurun_out/r04_reselect_parity.log | head -10 | cut -c1-300; exit 1; fi
timeout 300 python bench.py --dist docruns --no_cpu_baseline --no_traffic --no_also --steps 8 --warmup 4 --per_step > gpurun_out/r04_docruns_reselect.log 2>&1; echo "exit $?"
grep "^step" gpurun_out/r04_docruns_reselect.log | cut -c1-260
tail -1 gpurun_out/r04_docruns_reselect.log > gpurun_out/r04_bench_170M_b64_docruns.json
python -c "import json; d=json.load(open('gpurun_out/r04_bench_170M_b64_docruns.json')); print('  Q/s %.0f ms/step %.3f scan %.3f recall %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('recall_at_10')))"
timeout 300 python bench.py --no_cpu_baseline --no_traffic --no_also --steps 10 > gpurun_out/r04_b64_reselect.log 2>&1; echo "exit $?"
tail -1 gpurun_out/r04_b64_reselect.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  iid Q/s %.0f ms/step %.3f scan %.3f fixed %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['fixed_ms_per_step']))"
