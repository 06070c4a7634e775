#!/usr/bin/env python3
"""Debug aid: mixture dump at N rows -- libdph top-k vs an independent fp64 brute force in torch, planted-row ranks."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from densephrases_amd import Shard
from densephrases_amd.synth import synthetic_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
dev = torch.device("cuda", 0)
s = Shard(n, device=0)
s.fill_synthetic(seed=42, kind=1)
s.finalize()
print("shard stats", s.shard_stats())
rng = np.random.default_rng(1234)
B = 32
p = rng.integers(0, n, B)
rows = np.stack([synthetic_rows(int(r), 1, 42, 1)[0] for r in p]).astype(np.float32) / 20 - 2
x = (rows + rng.normal(0, 0.1, rows.shape)).astype(np.float32)
D, I = s.search(x, 10)
print("stats", s.stats())
xs = torch.from_numpy(x).to(dev)
ref_s, ref_i = bench.independent_topk(s.rows_dev_ptr(), n, 0, xs, 10, dev)
ref_i = ref_i.cpu().numpy(); ref_s = ref_s.cpu().numpy()
same = (ref_i == I).all(1)
print("rows identical to the brute force:", int(same.sum()), "/", B)
for r in range(B):
    rank_ref = np.nonzero(ref_i[r] == p[r])[0]
    rank_got = np.nonzero(I[r] == p[r])[0]
    if not same[r] or rank_ref.size == 0:
        print(r, "planted", p[r], "rank ref", rank_ref, "rank got", rank_got)
        print("   ref", ref_i[r][:6], np.round(ref_s[r][:6], 2))
        print("   got", I[r][:6], np.round(D[r][:6], 2))
