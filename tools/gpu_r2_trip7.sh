#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_scoring.py tests/test_model_facade.py tests/test_dist_nccl.py -m gpu -q --timeout 300 -p no:cacheprovider -k "merged or sharded or scoring or logits or facade or golden or from_reference or rccl" > gpurun_out/t7_pytest.log 2>&1
echo "exit $?"; tail -40 gpurun_out/t7_pytest.log
run() { name=$1; shift; timeout 400 python bench.py --steps 10 --warmup 3 --no_cpu_baseline "$@" > gpurun_out/t7_$name.log 2>&1; echo "$name exit $?"; tail -1 gpurun_out/t7_$name.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.readline()); r=d['roofline']
    print('   Q/s %.0f  ms/step %.2f  scan %.2f ms  hbm %.3f  batch-hbm %.3f  mfma %.3f  pairs %d trig %d fast %s recall %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['per_batch']['frac'], r['mfma_int8']['frac'], d['scan_pairs_last_launch'], d['scan_emit_triggers_last_launch'], d['certified_by_first_attempt_last_step'], d.get('recall_at_10', d.get('recall_error'))))
except Exception as e: print('   parse failed', e)
"; }
run b128 --batch 128
run b512 --batch 512 --steps 4 --warmup 2
run b128_mix --batch 128 --dist mixture
