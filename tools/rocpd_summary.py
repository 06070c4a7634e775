#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite result: per-kernel calls / total / average duration (us), the
equivalent of `--stats` kernel_stats.csv, plus launch geometry of the scan kernel.  Usage: rocpd_summary.py <db> [out]"""
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
cur = con.cursor()
lines = ["name,calls,total_us,avg_us,percent"]
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    lines.append(f"\"{name[:110]}\",{calls},{total / 1e3 if total > 1e7 else total:.3f},{avg / 1e3 if avg > 1e6 else avg:.3f},{pct:.3f}")
lines.append("")
lines.append("# dispatches of dph_scan_kernel: duration_us, grid, workgroup, lds, vgpr, agpr, sgpr")
for r in cur.execute("select duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels "
                     "where name like '%dph_scan_kernel<_, 4, %, 0, _, %>%' or name like '%dph_scan_units_kernel<0, %>%' order by start limit 40"):
    lines.append(",".join(str(x) for x in ((r[0] / 1e3,) + r[1:])))
lines.append("")
lines.append("# dispatches of the PQ chain's selection kernels, in order: name, duration_us, grid, workgroup")
for r in cur.execute("select name, duration, grid_x, workgroup_x from kernels where name like 'dph_coarse_select_kernel%' or name like 'dph_coarse_estimate_sample%' "
                     "or name like 'pq_final_kernel%' or name like 'dph_cf_flatten%' order by start limit 48"):
    lines.append(f"{r[0][:40]},{r[1] / 1e3:.1f},{r[2]},{r[3]}")
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
else:
    print(out)
