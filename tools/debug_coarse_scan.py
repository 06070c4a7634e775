#!/usr/bin/env python3
"""The candidate pool of the coarse quantizer's filter, variant 3 (GEMM) against variant 5 (filter scan), on a small synthetic PQ index:
both must hold every (query row, list) whose one-product bf16 score reaches the row's estimate -- same lists, scores equal up to the
fp32 summation order.  python tools/debug_coarse_scan.py [--nlist 65536] [--rows 5]"""
import argparse
import os

os.environ.setdefault("DPH_CF_KEEP_POOL", "1")      # the scan form writes the linear pool only on request (it buckets straight from its chunks)
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nlist", type=int, default=65536)
    ap.add_argument("--rows", type=int, default=5)
    a = ap.parse_args()
    import torch
    from densephrases_amd.synth import synthetic_pq_shard
    s, A, cent, sizes = synthetic_pq_shard(2_000_000, a.nlist, 96, device=0)
    x = np.random.default_rng(3).normal(0, 0.5, (a.rows, 768)).astype(np.float32)
    pools = {}
    for v in (3, 5):
        s.set_tuning("coarse_filter", v)
        D, I = s.search_ivf(x, 10, 256)
        failed, emitted = s.debug_pq_coarse()
        lists, score, q = s.debug_pq_pool()
        pools[v] = (lists, score, q, D, I)
        print(f"variant {v}: failed_over {failed} emitted {emitted} pool {lists.size} per row {np.bincount(q, minlength=a.rows)[:a.rows]}")
    l3, s3, q3, D3, I3 = pools[3]
    l5, s5, q5, D5, I5 = pools[5]
    print("ids equal", np.array_equal(I3, I5))
    xr = (x @ A.T).astype(np.float32)
    for r in range(min(a.rows, 3)):
        a3 = dict(zip(l3[q3 == r].tolist(), s3[q3 == r].tolist()))
        a5 = dict(zip(l5[q5 == r].tolist(), s5[q5 == r].tolist()))
        only3, only5 = sorted(set(a3) - set(a5)), sorted(set(a5) - set(a3))
        both = sorted(set(a3) & set(a5))
        d = np.array([a3[l] - a5[l] for l in both])
        dup5 = l5[q5 == r].size - len(a5)
        print(f"row {r}: |3| {len(a3)} |5| {len(a5)} (+{dup5} duplicates) common {len(both)} only3 {len(only3)} only5 {len(only5)} max|score diff| {np.abs(d).max() if d.size else 0:.3e}")
        for name, only, pool in (("only3", only3, a3), ("only5", only5, a5)):
            for l in only[:4]:
                exact = float(xr[r].astype(np.float64) @ cent[l].astype(np.float64))
                print(f"    {name}: list {l} (tile {l // 32}, row {l % 32}) pool score {pool[l]:.5f} exact {exact:.5f}")
        if a3:
            print(f"    min score in 3: {min(a3.values()):.5f}, in 5: {min(a5.values()) if a5 else None}")


if __name__ == "__main__":
    main()
