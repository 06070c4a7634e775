#!/usr/bin/env python3
"""Throughput of the IVF search path at BASELINE config-4 shape (IVF-4096, nprobe 256, batch 256) on one GPU.
Timing only: the shard is the synthetic dump with a RANDOM tile->list map and random centroids (the cost of the
coarse quantizer, the mask plumbing and the masked scan does not depend on what the lists mean; parity is covered by
tests/test_ivf.py on real k-means lists).  Prints one JSON line with the IVF unit scan, the IVF masked scan and the exact (flat) search side by side."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=170_000_000)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=256)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--skew", type=int, default=0, help="> 0: the queries come from this many clusters (each around a centroid): "
                    "hot lists probed by many query rows, several 128-slot chunks per list")
    ap.add_argument("--tune", action="append", default=[], help="key=value tuning pairs (dph_index_set_tuning)")
    ap.add_argument("--only", default="", help="comma list of ivf_units,ivf_masked,exact (default: all)")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import Shard
    n = args.rows // 32 * 32
    dev = torch.device("cuda", 0)
    s = Shard(n, device=0)
    s.fill_synthetic(seed=42)
    rng = np.random.default_rng(0)
    s.set_row_ids(np.arange(n, dtype=np.int64), n)
    # lists as contiguous runs of tiles (like a real list-major shard), random lengths
    cuts = np.sort(rng.choice(np.arange(1, n // 32), args.nlist - 1, replace=False))
    tile_list = np.zeros(n // 32, dtype=np.int32)
    tile_list[cuts] = 1
    tile_list = np.cumsum(tile_list).astype(np.int32)
    cent = rng.normal(0, 0.5, (args.nlist, 768)).astype(np.float32)
    s.set_ivf(cent, tile_list)
    s.finalize()
    for kv in args.tune:
        key, v = kv.split("=")
        s.set_tuning(key, int(v))
    R, k = 2 * args.batch, 10
    xq = rng.normal(0, 0.5, (R, 768)).astype(np.float32)
    if args.skew > 0:
        hot = rng.choice(args.nlist, args.skew, replace=False)
        xq = (cent[hot[rng.integers(0, args.skew, R)]] + 0.3 * xq).astype(np.float32)
    x = torch.from_numpy(xq).to(dev)
    D = torch.empty((R, k), dtype=torch.float32, device=dev)
    I = torch.empty((R, k), dtype=torch.int64, device=dev)
    st = torch.empty(R, dtype=torch.int32, device=dev)
    out = {}
    ivf = lambda: s.search_ivf_dev(x.data_ptr(), R, k, args.nprobe, D.data_ptr(), I.data_ptr(), st.data_ptr())
    ids = {}
    for name, units, fn in (("ivf_units", 1, ivf), ("ivf_masked", 0, ivf),
                            ("exact", 0, lambda: s.search_dev(x.data_ptr(), R, k, D.data_ptr(), I.data_ptr(), st.data_ptr()))):
        if args.only and name not in args.only.split(","):
            continue
        s.set_tuning("ivf_units", units)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s.profile_enable(True)
        s.profile_read()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        scan_ms, scan_n = s.profile_read()
        s.profile_enable(False)
        out[name] = {"ms_per_batch": dt * 1e3, "full_scan_ms_per_batch": scan_ms / args.steps, "full_scans_per_batch": scan_n / args.steps, "queries_per_sec": args.batch / dt, "certified": int((st == 0).sum().item())}
        ids[name] = I.clone()
    if "ivf_units" in ids and "ivf_masked" in ids:
        out["units_equals_masked"] = bool((ids["ivf_units"] == ids["ivf_masked"]).all().item())
    if "ivf_units" in ids:
        out["units_queue"] = s.debug_units()
    print(json.dumps({"rows": n, "nlist": args.nlist, "nprobe": args.nprobe, "batch": args.batch, "query_rows": R, "skew": args.skew, **out}))


if __name__ == "__main__":
    main()
