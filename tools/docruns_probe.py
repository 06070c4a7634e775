#!/usr/bin/env python3
"""Which rows of the bench's document-ordered batches does the first attempt leave open, and why?  (170 M rows of kind 2, the four
batches bench.py cycles, first attempt only: tuning key retry_chain = 0; bucket counts of the pass.)"""
import json
import sys
import os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import Shard
    from densephrases_amd.synth import synthetic_rows
    n, B, k, kind, seed = 170_000_000, 64, 10, 2, 42
    dev = torch.device("cuda", 0)
    s = Shard(n, device=0)
    s.fill_synthetic(seed=seed, kind=kind)
    s.finalize()
    rng = np.random.default_rng(1234)
    out = []
    for b in range(4):
        q = rng.normal(0, 0.5, (B, 1536)).astype(np.float32)
        p = rng.integers(0, n, B // 2)
        rows = np.stack([synthetic_rows(int(r), 1, seed, kind)[0] for r in p]).astype(np.float32) / 20 - 2
        q[:B // 2, :768] = rows + rng.normal(0, 0.1, rows.shape).astype(np.float32)
        x = torch.from_numpy(np.concatenate([q[:, :768], q[:, 768:]], 0)).to(dev)
        D = torch.empty((2 * B, k), dtype=torch.float32, device=dev)
        I = torch.empty((2 * B, k), dtype=torch.int64, device=dev)
        st = torch.empty(2 * B, dtype=torch.int32, device=dev)
        s.set_tuning("retry_chain", 0)
        s.search_dev(x.data_ptr(), 2 * B, k, D.data_ptr(), I.data_ptr(), st.data_ptr())
        torch.cuda.synchronize()
        raw, ov = s.debug_bucket_counts(2 * B)
        bad = np.nonzero(st.cpu().numpy() != 0)[0]
        out.append({"batch": b, "open_rows": bad.tolist(), "their_bucket_counts": raw[bad].tolist(), "pool_overflow": ov[bad].tolist(),
                    "bucket_count_max": int(raw.max()), "bucket_count_median": float(np.median(raw)), "stats": s.stats()})
        s.set_tuning("retry_chain", 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
