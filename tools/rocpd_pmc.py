#!/usr/bin/env python3
"""Per-kernel average of ONE PMC counter from a rocprofv3 (ROCm 7.2) rocpd SQLite result, plus the gfx950 FETCH_SIZE
correction (x2: the counter prices 128-B requests at 64 B, MI355X_MICROARCH.md HBM section).
Usage: rocpd_pmc.py <db> <out.csv> [<pmc_latest.json> <rows_per_gpu>]"""
import json
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                        "group by kernel_name, counter_name order by kernel_name"))
lines = ["kernel,counter,avg_value_KB,dispatches,hbm_bytes_corrected(x2*1024)"]
scan = None
for name, ctr, avg, n in rows:
    lines.append(f"\"{name[:90]}\",{ctr},{avg:.1f},{n},{int(avg * 2 * 1024)}")
    if "dph_scan_kernel<16, 24, false, true, false>" in name and ctr == "FETCH_SIZE":
        scan = (name, avg)
open(out, "w").write("\n".join(lines) + "\n")
if len(sys.argv) > 4 and scan:
    n_rows = int(sys.argv[4])
    B, k = 64, 10                       # bench.py defaults; SURVEY.md 8(d): N*d*1 + 2B*d*4 + 2B*k*12
    json.dump({"rows_per_gpu": n_rows, "kernel": "dph_scan_kernel<16, 24, false, true, false>", "counter": "FETCH_SIZE",
               "raw_value_kb_avg": scan[1],
               "gfx950_correction": "x2 (FETCH_SIZE counts 128-B requests at 64 B: MI355X_MICROARCH.md, HBM section)",
               "hbm_bytes_per_launch": int(scan[1] * 2 * 1024), "algorithmic_bytes_per_launch": n_rows * 768 + 2 * B * 768 * 4 + 2 * B * k * 12,
               "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --steps 3 --warmup 1 "
                         "(own pass, no other counters); tools/gpu_final.sh + tools/rocpd_pmc.py"},
              open(sys.argv[3], "w"), indent=1)
