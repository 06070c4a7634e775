#!/usr/bin/env python3
"""End-to-end MIPS.search throughput: the python class exactly as the reference's callers use it
(host numpy queries in, list-of-dicts out, aggregate=True), i.e. including PCIe copies, metadata lookup, dict
assembly, paragraph cropping and de-duplication.  Synthetic dump generated on the device; documents of 100 rows with
synthetic text.  Follows run_demo.py:329-352: fixed batch, first 5 batches excluded.
Usage: python tools/e2e_mips.py [--rows N] [--batch B] [--steps K]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from densephrases_amd.synth import SynthDocStore as SynthStore      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=170_000_000)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--warmup", type=int, default=5)
    args = ap.parse_args()
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import MIPS, Shard
    from densephrases_amd.synth import synthetic_rows
    n = args.rows
    shard = Shard(n, device=0)
    shard.fill_synthetic(seed=42)
    shard.set_idx2id((np.arange(n, dtype=np.int64) // 100).astype(np.int32), (np.arange(n, dtype=np.int64) % 100).astype(np.int32))
    nd = (n + 99) // 100
    shard.set_f2o(np.arange(nd, dtype=np.int32), np.arange(0, (nd + 1) * 100, 100, dtype=np.int64),
                  np.tile(np.arange(100, dtype=np.int32), nd))
    shard.finalize()
    mips = MIPS.from_shard(shard, SynthStore())
    rng = np.random.default_rng(5)
    B = args.batch
    batches = []
    for _ in range(4):
        q = rng.normal(0, 0.5, (B, 1536)).astype(np.float32)
        p = rng.integers(0, n - 8, B)
        rows = np.stack([synthetic_rows(int(r), 1, 42)[0] for r in p]).astype(np.float32) / 20 - 2
        rows_e = np.stack([synthetic_rows(int(r) + 2, 1, 42)[0] for r in p]).astype(np.float32) / 20 - 2
        q[:, :768] = rows + rng.normal(0, 0.1, rows.shape)
        q[:, 768:] = rows_e + rng.normal(0, 0.1, rows.shape)
        batches.append((q.astype(np.float64), p))
    for i in range(args.warmup):
        mips.search(batches[i % 4][0], q_texts=["q"] * B, top_k=10, aggregate=True, agg_strat="opt1")
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = mips.search(batches[i % 4][0], q_texts=["q"] * B, top_k=10, aggregate=True, agg_strat="opt1")
    dt = time.perf_counter() - t0
    # the same batches through search_stream: the GPU half of batch t+1 overlaps the host half of batch t
    for _ in mips.search_stream((batches[i % 4][0] for i in range(3)), top_k=10, aggregate=True):
        pass                                       # allocates the two record slots / pinned buffers
    t0 = time.perf_counter()
    n_out = 0
    for outs in mips.search_stream((batches[i % 4][0] for i in range(args.steps)), q_texts=(["q"] * B for _ in range(args.steps)),
                                   top_k=10, aggregate=True, agg_strat="opt1"):
        n_out += len(outs)
    dt_stream = time.perf_counter() - t0
    assert n_out == args.steps * B and [r[0]["start_idx"] for r in outs if r] == [r[0]["start_idx"] for r in out if r]
    q, p = batches[(args.steps - 1) % 4]
    ok = sum(1 for r, pr in zip(out, p) if r and r[0]["doc_idx"] == pr // 100 and r[0]["start_idx"] == pr % 100 and r[0]["end_idx"] == pr % 100 + 2)
    print(json.dumps({"metric": "end-to-end MIPS.search queries/sec (host in, dicts out)", "value": args.steps * B / dt,
                      "ms_per_batch": dt / args.steps * 1e3,
                      "search_stream_value": args.steps * B / dt_stream, "search_stream_ms_per_batch": dt_stream / args.steps * 1e3,
                      "rows": n, "batch": B, "top_k": 10,
                      "top1_is_planted_phrase": f"{ok}/{B}", "stats_last": shard.stats()}))


if __name__ == "__main__":
    main()
