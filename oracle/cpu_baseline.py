#!/usr/bin/env python3
"""The timed CPU comparator of bench.py (`cpu_baseline`), in a process of its own (numpy only: with torch loaded into the
same process numpy's BLAS ran 6x slower on the build container -- two OpenMP runtimes on the same cores).

The FAISS-CPU IndexFlatIP execution shape on a bounded sample, all host cores: fp32 vectors resident in RAM (de-quantised
once, like an index built from the dump), one sgemm per block on the host BLAS, running top-k
(oracle.mips_oracle.flat_ip_search_fp32_resident).  The sample has DISTINCT rows of the dump's distribution
(int8 codes ~ 40 + 12 z, the i.i.d. dump of BASELINE config 2).  Prints one JSON object.
Usage: python -m oracle.cpu_baseline --batch 64 --top_k 10 --rows 393216 --budget 12"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.mips_oracle import flat_ip_search_fp32_resident      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--top_k", type=int, default=10)
    ap.add_argument("--rows", type=int, default=393216)
    ap.add_argument("--budget", type=float, default=12.0, help="seconds of timed passes")
    a = ap.parse_args()
    rng = np.random.default_rng(7)
    blk = 16384
    n_cpu = max(blk, a.rows // blk * blk)
    blocks = []
    for _ in range(n_cpu // blk):
        nb = np.clip(np.rint(40.0 + 12.0 * rng.standard_normal((blk, 768), dtype=np.float32)), -128, 127)
        blocks.append((nb / 20.0 - 2.0).astype(np.float32))                  # x = n/20 - 2 (embed_utils.py:148)
    q = rng.normal(0, 0.5, (2 * a.batch, 768)).astype(np.float32)
    flat_ip_search_fp32_resident(q, blocks[:2], a.top_k)                     # warm-up (BLAS thread pool)
    times, t_start = [], time.time()
    while len(times) < 3 or (time.time() - t_start < a.budget and len(times) < 400):
        t0 = time.time()
        flat_ip_search_fp32_resident(q, blocks, a.top_k)
        times.append(time.time() - t0)
    t = float(np.median(times))
    print(json.dumps({"rows": n_cpu, "block": blk, "seconds_per_pass": t, "passes": len(times), "cores": os.cpu_count() or 1,
                      "qps_sample": a.batch / t, "gflops": 2 * (2 * a.batch) * 768 * n_cpu / t / 1e9}))


if __name__ == "__main__":
    main()
