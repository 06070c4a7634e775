#!/usr/bin/env python3
"""The strong-scaling bench (170 M rows over W ranks) emulated on ONE MI355X: W resident shards with their ranks' id
ranges (together the same synthetic dump as bench.py), every step runs each emulated rank's phases back to back with
the collectives replaced by device copies, then the real W-way record merge.  Reports, for the one-phase search (every
shard under its own bound) and the two-phase search (union bound), the slowest rank's time per phase and the step time
an W-GPU node would see if the two small all-gathers were free; checks planted rows + certificates of the merged result.
Usage: python tools/scale_emulated.py [--world 8] [--rows 170000000] [--steps 6]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rows", type=int, default=170_000_000)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import Shard
    from densephrases_amd.dist import RecordLayout, ShardedSearcher, exchange_and_merge, partition_rows
    from densephrases_amd.synth import synthetic_rows
    W, n_total, B, k, L = args.world, args.rows, args.batch, 10, 10
    dev = torch.device("cuda", 0)
    shards = []
    for lo, hi in partition_rows(n_total, W):
        n = hi - lo
        s = Shard(n, device=0, id_base=lo)
        s.fill_synthetic(seed=42)
        s.set_idx2id(((np.arange(n, dtype=np.int64) + lo) // 100).astype(np.int32),
                     ((np.arange(n, dtype=np.int64) + lo) % 100).astype(np.int32))
        d0, d1 = lo // 100, (hi + 99) // 100
        ids = np.arange(d0, d1 + 1, dtype=np.int32)
        s.set_f2o(ids, np.arange(0, (len(ids) + 1) * 100, 100, dtype=np.int64), np.tile(np.arange(100, dtype=np.int32), len(ids)))
        s.finalize()
        shards.append(s)
    rng = np.random.default_rng(1234)
    batches, planted = [], []
    for _ in range(4):
        q = rng.normal(0, 0.5, (B, 1536)).astype(np.float32)
        p = rng.integers(0, n_total, B // 2)
        rows = np.stack([synthetic_rows(int(r), 1, 42)[0] for r in p]).astype(np.float32) / 20 - 2
        q[:B // 2, :768] = rows + rng.normal(0, 0.1, rows.shape).astype(np.float32)
        batches.append(torch.from_numpy(q).to(dev))
        planted.append(p)
    layout = RecordLayout(2 * B, k)
    rec_all = torch.zeros((W, layout.nbytes), dtype=torch.uint8, device=dev)
    top_all = torch.empty((W, 2 * B, 16), dtype=torch.int32, device=dev)

    class NoDist:
        @staticmethod
        def all_gather_into_tensor(out, inp):
            pass

    def ev():
        return torch.cuda.Event(enable_timing=True)

    report = {}
    for mode in ("own_bound", "union_bound"):
        ss = [ShardedSearcher(s, B, k, L, device=dev, union_bounds=(mode == "union_bound")) for s in shards]
        for x in ss:                       # what ShardedSearcher does itself when it is constructed with world = W
            if mode == "union_bound":
                x.set_sample_world(W)
            else:
                x.shard.set_tuning("fine_stride", 0)
        merger = ss[0]
        t_a, t_b, t_m, uncert = [], [], [], 0
        for it in range(args.steps + 2):
            q = batches[it % 4]
            ta, tb = [], []
            for r, s in enumerate(ss):
                e0, e1 = ev(), ev()
                e0.record()
                s.load_query(q)
                if s.union_bounds:
                    s.sample()
                e1.record()
                torch.cuda.synchronize()
                ta.append(e0.elapsed_time(e1))
                if s.union_bounds:
                    top_all[r].copy_(s.top)
            for r, s in enumerate(ss):
                e0, e1 = ev(), ev()
                e0.record()
                if s.union_bounds:
                    s.union_bound(top_all, W)
                s.search_and_rescore()
                e1.record()
                torch.cuda.synchronize()
                tb.append(e0.elapsed_time(e1))
                rec_all[r].copy_(s.rec)
            e0, e1 = ev(), ev()
            merger.world = W
            e0.record()
            D, I, best, pred, status = exchange_and_merge(layout, merger.rec, rec_all, NoDist, W, merger._merge)
            e1.record()
            torch.cuda.synchronize()
            merger.world = 1
            assert (I[:B // 2, 0].cpu().numpy() == planted[it % 4]).all(), f"{mode}: planted rows did not come back first"
            if it >= 2:
                t_a.append(max(ta)); t_b.append(max(tb)); t_m.append(e0.elapsed_time(e1))
                uncert += int((status != 0).sum())
        if mode == "own_bound":
            ref = (D.clone(), I.clone(), best.clone(), pred.clone())
        else:
            same = all(bool((a == b).all()) for a, b in zip(ref, (D, I, best, pred)))
            report["union_equals_own_last_batch"] = same
            assert same, "two-phase result differs from the one-phase result"
        a, b, m = float(np.mean(t_a)), float(np.mean(t_b)), float(np.mean(t_m))
        report[mode] = {"slowest_rank_sample_ms": a, "slowest_rank_search_rescore_ms": b, "merge_ms": m,
                        "step_ms_if_allgathers_free": a + b + m, "queries_per_sec": B / ((a + b + m) / 1e3),
                        "uncertified_rows": uncert}
    report.update({"world": W, "rows_total": n_total, "batch": B})
    print(json.dumps(report))


if __name__ == "__main__":
    main()
