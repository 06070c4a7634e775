#!/bin/bash
# round-2 trip 1: first contact of the filter-scan pipeline with the hardware
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== stage 1: bucket tests (scan + refine exactness)"
timeout 300 python -m pytest tests/test_gpu_search.py -m gpu -q --timeout 120 -p no:cacheprovider -x -k "scan_buckets" > gpurun_out/t1_buckets.log 2>&1
rc=$?; echo "exit $rc"; tail -25 gpurun_out/t1_buckets.log
if [ $rc -eq 124 ]; then echo "stage 1 timed out: stopping"; exit 1; fi
echo "== stage 2: search parity"
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_ivf.py -m gpu -q --timeout 200 -p no:cacheprovider -k "not full_size and not scan_buckets" > gpurun_out/t1_pytest.log 2>&1
rc=$?; echo "exit $rc"; tail -40 gpurun_out/t1_pytest.log
if [ $rc -eq 124 ]; then echo "stage 2 timed out: stopping"; exit 1; fi
echo "== stage 3: bench 170M"
run() { name=$1; shift; timeout 400 python bench.py --steps 10 --warmup 3 --no_cpu_baseline "$@" > gpurun_out/t1_$name.log 2>&1; echo "$name exit $?"; tail -2 gpurun_out/t1_$name.log | cut -c1-1500; }
run b64_n8
run b64_n4 --tune scan_nset_qb1=4
run b128_n4 --batch 128
run b128_n6 --batch 128 --tune scan_nset_qb2=6
run b256 --batch 256 --steps 6
run b64_mix --dist mixture
run b64_kp8 --tune sample_kp=8
