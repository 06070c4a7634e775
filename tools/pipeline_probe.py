#!/usr/bin/env python3
"""Two batches in flight (dist.PipelinedSearcher) against the plain step on the configs[1] shard: queries/sec for several splits of
the chip (CUs of the side stream) and finest sampled strides (the side stream has to get through 1/stride of the dump while the main
stream scans the rest).  Every configuration's last result is compared with the plain step's.  Prints one JSON line.
Usage: python tools/pipeline_probe.py [--rows 170000000] [--batch 64] [--steps 12] [--configs 8:32,8:48,8:64,16:32]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=170_000_000)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--configs", default="8:32,8:48,8:64,16:32,16:24")
    a = ap.parse_args()
    import numpy as np
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.dist import PipelinedSearcher, ShardedSearcher
    from densephrases_amd.synth import synthetic_queries, synthetic_rows
    dev = torch.device("cuda", 0)
    B, k, L = a.batch, 10, 10
    s = Shard(a.rows, device=0)
    s.fill_synthetic(seed=42, kind=0)
    n_docs = a.rows // 100
    s.set_idx2id((np.arange(a.rows, dtype=np.int64) // 100).astype(np.int32), (np.arange(a.rows, dtype=np.int64) % 100).astype(np.int32))
    s.set_f2o(np.arange(n_docs, dtype=np.int32), np.arange(0, n_docs * 100 + 1, 100, dtype=np.int64), np.tile(np.arange(100, dtype=np.int32), n_docs))
    s.finalize()
    rng = np.random.default_rng(1)
    n_b = a.steps + a.warmup
    qs = []
    for i in range(n_b):
        rows = rng.integers(0, a.rows, 2 * B)
        planted = np.concatenate([synthetic_rows(int(r), 1, 42, 0) for r in rows])
        x = synthetic_queries(2 * B, seed=100 + i, planted_rows=planted)
        qs.append(torch.from_numpy(np.concatenate([x[:B], x[B:]], 1)).to(dev))

    def timed(step, flush=None):
        outs = None
        for q in qs[:a.warmup]:
            step(q)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for q in qs[a.warmup:]:
            outs = step(q)
        if flush is not None:
            outs = flush()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt / a.steps * 1e3, {key: outs[key].cpu().numpy().copy() for key in ("D", "I", "best", "pred", "status")}

    plain = ShardedSearcher(s, B, k, L, device=dev)
    ms0, want = timed(plain.step)
    ms0b, _ = timed(plain.step)
    res = {"rows": a.rows, "batch": B, "steps": a.steps, "plain_ms": [ms0, ms0b], "plain_qps": B / min(ms0, ms0b) * 1e3, "configs": []}
    print("plain", ms0, ms0b, file=sys.stderr, flush=True)
    for c in a.configs.split(","):
        side, fine = (int(v) for v in c.split(":"))
        pipe = PipelinedSearcher(s, B, k, L, side_cus=side)
        for lane in pipe.lanes:
            lane.shard.set_tuning("fine_stride", fine)
        ms, got = timed(pipe.step, pipe.flush)
        ms2, got = timed(pipe.step, pipe.flush)
        same = all(bool((got[key] == want[key]).all()) for key in got)
        st = pipe.lanes[0].shard.stats()
        r = {"side_cus": side, "fine_stride": fine, "ms_per_batch": [ms, ms2], "qps": B / min(ms, ms2) * 1e3, "same_as_plain": same,
             "speedup": min(ms0, ms0b) / min(ms, ms2), "uncertified": st["uncertified"], "certified_fast": st["certified_fast"]}
        res["configs"].append(r)
        print(r, file=sys.stderr, flush=True)
        pipe.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
