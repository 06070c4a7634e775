#!/usr/bin/env python3
"""How many CUs does the flat scan need?  The full scan is a persistent kernel with one workgroup per CU (256); the 128-row pass is
HBM-bound, so fewer workgroups might stream the dump just as fast -- and the CUs left over could run the latency-bound chain of the
NEXT batch (ladder, refine, threshold, select) on a second stream instead of ahead of the scan (the fixed 1.6 ms per step).
Times dph_debug_scan_time (the scan alone, nothing emitted) at 128 and 256 query rows for several grids (tuning key scan_grid; one
shard per grid: scratch is sized by it).  Prints one JSON line.  Usage: python tools/scan_grid_probe.py [--rows 170000000]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=170_000_000)
    ap.add_argument("--grids", default="256,248,240,224,192,128")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    import numpy as np
    from densephrases_amd import Shard
    x = np.random.default_rng(0).normal(0, 0.5, (256, 768)).astype(np.float32)
    out = {}
    for g in [int(v) for v in a.grids.split(",")]:
        s = Shard(a.rows, device=0)
        s.fill_synthetic(seed=42, kind=0)
        s.set_tuning("scan_grid", g)
        s.finalize()
        r = {}
        for n_q in (128, 256):
            ms = [float(v) for v in s.debug_scan_time(x[:n_q], a.iters)]
            r[str(n_q)] = {"ms": [round(v, 3) for v in ms], "min_ms": min(ms[1:]), "tbytes_per_s": a.rows * 768 / min(ms[1:]) / 1e9}
        out[str(g)] = r
        s.close()
        print(g, r, file=sys.stderr, flush=True)
    print(json.dumps({"rows": a.rows, "by_grid": out}))


if __name__ == "__main__":
    main()
