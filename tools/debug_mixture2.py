#!/usr/bin/env python3
"""Debug aid: which rows of the bench's mixture batches fail the first attempt, and why (bucket sizes)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from densephrases_amd import Shard
from densephrases_amd.synth import synthetic_rows
n = int(sys.argv[1]) if len(sys.argv) > 1 else 170_000_000
B = 64
s = Shard(n, device=0); s.fill_synthetic(seed=42, kind=1); s.finalize()
print(s.shard_stats())
rng = np.random.default_rng(1234)
for b in range(4):
    q = rng.normal(0, 0.5, (B, 1536)).astype(np.float32)
    p = rng.integers(0, n, B // 2)
    rows = np.stack([synthetic_rows(int(r), 1, 42, 1)[0] for r in p]).astype(np.float32) / 20 - 2
    q[:B // 2, :768] = rows + rng.normal(0, 0.1, rows.shape).astype(np.float32)
    x = np.concatenate([q[:, :768], q[:, 768:]], 0)
    D, I = s.search(x, 10)
    st = s.stats()
    print("batch", b, st, "pairs/triggers", s.scan_counters())
    if st["certified_fast"] < 2 * B:
        # find the rows: run them one by one
        for r in range(2 * B):
            s.search(x[r:r + 1], 10)
            t = s.stats()
            if t["certified_fast"] == 0:
                print("   row", r, t, "planted" if r < B // 2 else "random", "top scores", np.round(D[r][:4], 1), "ids", I[r][:3])
