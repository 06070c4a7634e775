#!/bin/bash
# Round-3 trip 4: diagnosis of the built-shard slowdown (random assignment: nothing heavy runs before the searches), clocks
# sampled in the background; the new bench line with its `also` legs.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( while true; do date +%s.%N; rocm-smi --showclocks --showpower --showtemp --showmemuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature|VRAM%|GPU Memory" ; sleep 1; done ) > gpurun_out/r03_t4_smi.log 2>&1 &
SMI=$!
echo "== probe (random assignment)"
timeout 500 python tools/ivf_slow_probe.py --assign random > gpurun_out/r03_t4_probe_random.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t4_probe_random.log
kill $SMI
echo "== bench with also legs"
timeout 600 python bench.py > gpurun_out/r03_t4_bench.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t4_bench.log > gpurun_out/r03_t4_bench.json; python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r03_t4_bench.json"))
    print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("recall_at_1"))
    print(json.dumps(d.get("also"), indent=None)[:3000])
    print(json.dumps(d.get("cpu_baseline"))[:1200])
except Exception as e:
    print("parse failed", e)
PY
tail -5 gpurun_out/r03_t4_bench.log | cut -c1-400
