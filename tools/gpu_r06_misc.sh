#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== torch CPU comparator alone"; nproc
timeout 300 python -m oracle.cpu_baseline_torch --gib 8 --budget 12 > gpurun_out/r06_cpu_torch.log 2>&1; echo "exit $?"; tail -6 gpurun_out/r06_cpu_torch.log | cut -c1-600
echo "== 256-row pass variants (s_setprio, 16x16x64)"
timeout 900 python tools/scan_diag.py --rows 170000000 --only 0 512 1024 63 1087 --out gpurun_out/r06_scan_diag_256rows_variants.json > gpurun_out/r06_scan_diag.log 2>&1; echo "exit $?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06_scan_diag_256rows_variants.json"))
    for v in d["variants"]: print("  %5d %-70s %s" % (v["bits"], v["variant"], ("%.2f ms  %.0f TOP/s" % (v["median_ms_after_first"], v["int8_top_s"])) if "ms" in v else v.get("error", "")[-200:]))
except Exception as e: print("parse failed", e)
PY
