#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest -m gpu (all but the full-size test)"
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "not full_size" > gpurun_out/t6_pytest.log 2>&1
echo "exit $?"; tail -30 gpurun_out/t6_pytest.log
run() { name=$1; shift; timeout 400 python bench.py --steps 10 --warmup 3 --no_cpu_baseline "$@" > gpurun_out/t6_$name.log 2>&1; echo "$name exit $?"; tail -1 gpurun_out/t6_$name.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.readline()); r=d['roofline']
    print('   Q/s %.0f  ms/step %.2f  scan %.2f ms  hbm %.3f  batch-hbm %.3f  mfma %.3f  pairs %d trig %d fast %s recall %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['per_batch']['frac'], r['mfma_int8']['frac'], d['scan_pairs_last_launch'], d['scan_emit_triggers_last_launch'], d['certified_by_first_attempt_last_step'], d.get('recall_at_10', d.get('recall_error'))))
except Exception as e: print('   parse failed', e)
"; }
run b64_mix --dist mixture
run b64
run b128
