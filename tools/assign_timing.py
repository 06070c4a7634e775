#!/usr/bin/env python3
"""Throughput of the fused list assignment (dph_index_assign_dev) over the rows of a resident shard: synthetic dump,
random centroids.  Prints one JSON line: rows/s, the MFMA rate it implies, the projected time for 170 M rows."""
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
    ap.add_argument("--rows", type=int, default=8_000_000)
    ap.add_argument("--nlist", type=int, default=4096)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import Shard
    dev = torch.device("cuda", 0)
    n = args.rows // 32 * 32
    s = Shard(n, device=0)
    s.fill_synthetic(seed=7)
    rng = np.random.default_rng(0)
    c = torch.from_numpy(rng.normal(0, 0.5, (args.nlist, 768)).astype(np.float32)).to(dev)
    best = torch.empty(n, dtype=torch.int32, device=dev)
    gap = torch.empty(n, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    s.assign_lists_dev(c.data_ptr(), args.nlist, best.data_ptr(), gap.data_ptr(), n=min(n, 1 << 20), stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.assign_lists_dev(c.data_ptr(), args.nlist, best.data_ptr(), gap.data_ptr(), stream=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flop = 2.0 * n * args.nlist * 768
    counts = torch.bincount(best.to(torch.int64), minlength=args.nlist)
    print(json.dumps({"rows": n, "nlist": args.nlist, "seconds": dt, "rows_per_sec": n / dt, "tflops_fp32_mfma": flop / dt / 1e12,
                      "projected_seconds_170M_rows": 170e6 / (n / dt), "lists_used": int((counts > 0).sum().item()),
                      "largest_list": int(counts.max().item())}))


if __name__ == "__main__":
    main()
