#!/bin/bash
# round-2 trip 3: the lean inner loop (asm MFMAs, static LDS offsets, unconditional feed)
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== stage 1: bucket tests"
timeout 300 python -m pytest tests/test_gpu_search.py -m gpu -q --timeout 120 -p no:cacheprovider -x -k "scan_buckets" > gpurun_out/t3_buckets.log 2>&1
rc=$?; echo "exit $rc"; tail -5 gpurun_out/t3_buckets.log
if [ $rc -ne 0 ]; then echo "stage 1 failed: stopping"; tail -40 gpurun_out/t3_buckets.log; exit 1; fi
echo "== stage 2: search parity (subset)"
timeout 600 python -m pytest tests/test_gpu_search.py tests/test_ivf.py -m gpu -q --timeout 200 -p no:cacheprovider -k "search_matches or ivf or union or duplicate or lost_pairs or mixture" > gpurun_out/t3_pytest.log 2>&1
rc=$?; echo "exit $rc"; tail -8 gpurun_out/t3_pytest.log
echo "== stage 3: bench 170M"
run() { name=$1; shift; timeout 400 python bench.py --steps 10 --warmup 3 --no_cpu_baseline "$@" > gpurun_out/t3_$name.log 2>&1; echo "$name exit $?"; tail -1 gpurun_out/t3_$name.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.readline()); r=d['roofline']
    print('   Q/s %.0f  ms/step %.2f  scan %.2f ms  hbm %.3f  batch-hbm %.3f  mfma %.3f  pairs %d trig %d fast %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['per_batch']['frac'], r['mfma_int8']['frac'], d['scan_pairs_last_launch'], d['scan_emit_triggers_last_launch'], d['certified_by_first_attempt_last_step']))
except Exception as e: print('   parse failed', e)
"; }
run b64_n8
run b64_n4 --tune scan_nset_qb1=4
run b64_f64 --tune fine_stride=64
run b64_f128 --tune fine_stride=128
run b128 --batch 128
run b128_f64 --batch 128 --tune fine_stride=64
run b256 --batch 256 --steps 6
run b64_mix --dist mixture
run b128_mix --dist mixture --batch 128
