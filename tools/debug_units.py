#!/usr/bin/env python3
"""Debug aid for the IVF unit scan: a small clustered shard, unit scan vs masked scan row by row, work-queue counters."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import __graft_entry__ as g
    g.build()
    from test_ivf import _clustered_db, _ivf_shard
    from densephrases_amd.ivf import train_centroids
    n_rows, nlist, nprobe, n_q, k = 30000, 64, 8, 130, 10
    rng = np.random.default_rng(n_rows + nlist)
    xb, centres = _clustered_db(rng, n_rows, 24)
    cent = train_centroids(xb, nlist, iters=5, seed=3)
    s, assign = _ivf_shard(xb, cent, id_base=500, units=1)
    x = (centres[rng.integers(0, 24, n_q)] + rng.normal(0, 0.3, (n_q, 768))).astype(np.float32)
    D1, I1 = s.search_ivf(x, k, nprobe)
    print("units  :", s.debug_units(), "scan counters", s.scan_counters(), "stats", s.stats())
    s.set_tuning("ivf_units", 0)
    D0, I0 = s.search_ivf(x, k, nprobe)
    print("masked : scan counters", s.scan_counters(), "stats", s.stats())
    bad = np.nonzero((I1 != I0).any(1))[0]
    print("rows differing:", len(bad), bad[:20])
    for r in bad[:3]:
        print(r, "units", I1[r], "masked", I0[r])


if __name__ == "__main__":
    main()
