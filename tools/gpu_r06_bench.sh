#!/bin/bash
# round 6: the default bench line (all legs) -> gpurun_out/r06_bench_170M_b64.{log,json} + a short summary
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
RND=${RND:-r06}; name=${1:-170M_b64}; shift || true
timeout 1500 python bench.py "$@" > gpurun_out/${RND}_bench_$name.log 2>&1; echo "$name exit $?"; tail -1 gpurun_out/${RND}_bench_$name.log > gpurun_out/${RND}_bench_$name.json
RND=$RND python - "$name" <<'PY'
import json,sys,os
try:
    d=json.load(open(f"gpurun_out/{os.environ['RND']}_bench_{sys.argv[1]}.json")); r=d["roofline"]
    print("   Q/s %.0f  ms/step %.3f  scan %.3f ms  hbm %.3f  traffic/alg %s  wall %.0f s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r.get("traffic_over_algorithmic"), d.get("bench_wall_seconds", 0)))
    for k in ("ms_per_step_median", "ms_per_step_min", "distinct_batches"): print("  ", k, d.get(k))
    for k, v in d.get("also", {}).items():
        print("   also.%s: %s Q/s, %s ms, leg %.1f s %s" % (k, v.get("queries_per_sec"), v.get("ms_per_batch"), v.get("leg_seconds", 0), v.get("error", "")))
        for kk in ("e2e_mips_search", "b512_document_stream", "giant", "b256"):
            if kk in v: print("        ." + kk + ": " + json.dumps(v[kk])[:700])
    if "cpu_baseline" in d: print("   cpu:", json.dumps({k: v for k, v in d["cpu_baseline"].items() if k != "sample"})[:400])
except Exception as e: print("   parse failed", e); os.system(f"tail -5 gpurun_out/{os.environ['RND']}_bench_{sys.argv[1]}.log")
PY
