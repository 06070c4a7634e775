#!/bin/bash
# round 4, GPU trip 2: staggered hand-over schedules of the scan kernel -- parity first, then timing
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== parity of the schedules"
timeout 900 python -m pytest tests/test_gpu_search.py -m gpu -q -x -k "staggered or scan_buckets or fused_finest" -p no:cacheprovider 2>&1 | tail -5
echo "== timing"
timeout 600 python tools/scan_diag.py --scheds --rows 170000000 --iters 6 --out gpurun_out/r04_scan_scheds.json 2> gpurun_out/r04_scan_scheds.log | tail -c 200
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_scan_scheds.json"))
for v in d["schedules"]:
    print("  n_q %3d sched %d rep %d  %.2f ms  int8 %.3f  hbm %.2f TB/s" % (v["n_q"], v["sched"], v["rep"], v["median_ms_after_first"], v["int8_frac_of_5000"], v["hbm_tb_s"]))
PY
echo "== bench b64 with sched 2 on both kernels (headline kernel check)"
timeout 600 python bench.py --tune scan_sched=2 --no_also --no_cpu_baseline --no_traffic --steps 10 > gpurun_out/r04_bench_b64_sched2.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r04_bench_b64_sched2.log | cut -c1-400
