#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 rocpd SQLite result: start (us, relative), duration, queue / stream, name -- for looking at what
overlaps what (tools/pipeline_probe.py).  Usage: rocpd_timeline.py <db> [out.csv] [--min_us 0] [--last N]"""
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
con = sqlite3.connect(args[0])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
extra = [c for c in cols if "queue" in c or "stream" in c]
sel = ", ".join(["name", "start", "duration"] + extra)
rows = list(cur.execute(f"select {sel} from kernels order by start"))
last = int(opts.get("--last", "400"))
rows = rows[-last:]
t0 = rows[0][1] if rows else 0
lines = ["start_us,dur_us," + ",".join(extra) + ",name"]
for r in rows:
    lines.append(f"{(r[1] - t0) / 1e3:.1f},{r[2] / 1e3:.1f}," + ",".join(str(v) for v in r[3:]) + f",\"{r[0][:70]}\"")
out = "\n".join(lines) + "\n"
if len(args) > 1:
    open(args[1], "w").write(out)
else:
    print(out)
