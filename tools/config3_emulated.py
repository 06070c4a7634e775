#!/usr/bin/env python3
"""SURVEY.md 8(d) config 3 on ONE MI355X: the 8-GPU range-sharded search (1.3 B rows -> 162.5 M rows = 125 GB int8 per
GPU), emulated as the guide prescribes -- one shard at the full per-GPU size, 8 shard passes (one per emulated rank,
each over the same resident rows with its rank's id range), then the real 8-way record merge (dph_merge_records_dev over
the packed exchange records, exactly the buffer an all_gather_into_tensor would deliver).  Reports the per-shard step
time, the merge time and the throughput an 8-GPU node would reach if the all-gather were free (it moves 8 x 30 KB).
Usage: python tools/config3_emulated.py [--rows_per_gpu N] [--steps K]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows_per_gpu", type=int, default=162_500_000)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import Shard, _lib
    from densephrases_amd.dist import ShardedSearcher, exchange_and_merge
    from densephrases_amd.synth import synthetic_rows
    n, W, B, k, L = args.rows_per_gpu, args.world, args.batch, 10, 10
    dev = torch.device("cuda", 0)
    shard = Shard(n, device=0)
    shard.fill_synthetic(seed=42)
    shard.set_idx2id((np.arange(n, dtype=np.int64) // 100).astype(np.int32), (np.arange(n, dtype=np.int64) % 100).astype(np.int32))
    nd = (n + 99) // 100
    shard.set_f2o(np.arange(nd, dtype=np.int32), np.arange(0, (nd + 1) * 100, 100, dtype=np.int64),
                  np.tile(np.arange(100, dtype=np.int32), nd))
    shard.finalize()
    ss = ShardedSearcher(shard, B, k, L, device=dev)
    lay = ss.layout
    rec_all = torch.zeros((W, lay.nbytes), dtype=torch.uint8, device=dev)
    merger = ShardedSearcher(shard, B, k, L, device=dev)
    merger.world = W                                # its fused merge (dph_merge_records_dev) reads W parts
    merge = merger._merge

    class NoDist:                                   # the gathered buffer is filled by the emulated ranks below
        @staticmethod
        def all_gather_into_tensor(out, rec):
            pass

    rng = np.random.default_rng(3)
    q = rng.normal(0, 0.5, (B, 1536)).astype(np.float32)
    p = rng.integers(0, n, B // 2)                  # planted in emulated rank 5's range
    rows = np.stack([synthetic_rows(int(r), 1, 42)[0] for r in p]).astype(np.float32) / 20 - 2
    q[:B // 2, :768] = rows + rng.normal(0, 0.1, rows.shape).astype(np.float32)
    qd = torch.from_numpy(q).to(dev)
    qnoise = torch.from_numpy(rng.normal(0, 0.5, (B, 1536)).astype(np.float32)).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t_pass, t_merge = [], []
    for it in range(args.steps + 1):
        for r in range(W):
            # rank 5 holds the rows the planted queries point at; the other ranks see the same rows as distractors
            # under their own id range (their planted scores are suppressed by searching unrelated queries)
            ev[0].record()
            ss.step(qd if r == 5 else qnoise)
            ev[1].record()
            ss.v["I"].add_(r * n)                   # local -> global ids of emulated rank r
            rec_all[r].copy_(ss.rec)
            torch.cuda.synchronize()
            if it:
                t_pass.append(ev[0].elapsed_time(ev[1]))
        ev[2].record()
        D, I, best, pred, status = exchange_and_merge(lay, ss.rec, rec_all, NoDist, W, merge)
        ev[3].record()
        torch.cuda.synchronize()
        if it:
            t_merge.append(ev[2].elapsed_time(ev[3]))
    I0 = I[:B // 2, 0].cpu().numpy()
    assert (I0 == p + 5 * n).all(), "merged top-1 is not the planted row of emulated rank 5"
    assert int((status != 0).sum()) == 0
    tp, tm = float(np.mean(t_pass)), float(np.mean(t_merge))
    print(json.dumps({"config": "configs[2] emulated on 1 GPU: 8 shard passes + 8-way merge", "rows_per_gpu": n,
                      "rows_total": n * W, "batch": B, "top_k": k, "shard_step_ms": tp,
                      "merge_and_follow_ms": tm, "merged_top1_is_planted": f"{len(I0)}/{len(I0)}",
                      "projected_8gpu_queries_per_sec_if_allgather_free": B / ((tp + tm) / 1e3),
                      "note": "all-gather of 8 x %d B not measurable on one GPU; the driver's SCALE run measures it" % lay.nbytes}))


if __name__ == "__main__":
    main()
