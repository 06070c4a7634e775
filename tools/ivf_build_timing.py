#!/usr/bin/env python3
"""Full-size IVF build where the rows lie, then IVF against exact search over the built shard (BASELINE configs[3]).

A synthetic flat shard resident in HBM (--kind 0 i.i.d. / 1 mixture of 4096 Gaussians + outliers / 2 document-ordered
runs / 3 mixture) -> centroids (--centroids kmeans: spherical k-means over a sample of the resident rows, HIP update;
random: N(0, 0.5^2)) -> list assignment (fused MFMA GEMM + arg-max over the int8 rows) -> device-side list builder (radix
sort + gather) -> [timings on the buffer the builder allocated next to the original] -> rows moved into a fresh
allocation (dph_index_rehome_rows) -> batches through the IVF search and the exact search over the same list-major shard.

Prints one JSON line: phase times, per search ms / full-scan ms / first-attempt certificates / pairs per scan wave
(the pair pool's load), and recall@1/5/10 of IVF against the exact search of the same run."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_queries(kind, n_rows, R, seed, how, noise=0.25):
    from densephrases_amd.synth import synthetic_rows
    rng = np.random.default_rng(seed)
    if how == "random":
        return rng.normal(0, 0.5, (R, 768)).astype(np.float32)
    # "near": a stored row, de-quantised, plus N(0, 0.25^2) -- on the mixture dump a fresh member of that row's cluster
    rows = rng.integers(0, n_rows, R)
    base = np.concatenate([synthetic_rows(int(r), 1, seed=42, kind=kind) for r in rows]).astype(np.float32) / 20.0 - 2.0
    return (base + rng.normal(0, noise, base.shape)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=170_000_000)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=256)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--kind", type=int, default=3)
    ap.add_argument("--centroids", default="kmeans", choices=["kmeans", "random"])
    ap.add_argument("--queries", default="near", choices=["near", "random"])
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--train_rows", type=int, default=0, help="0 = the trainer's default (4 %% of the rows, 39..256 per list)")
    ap.add_argument("--sweep", default="", help="after the main measurement: comma list of nprobe values; recall@1/5/10 against the exact "
                    "search and ms per batch for each, on queries = stored row + N(0, --sweep_noise^2)")
    ap.add_argument("--sweep_noise", type=float, default=0.5)
    ap.add_argument("--before_rehome", action="store_true", help="also time the searches on the buffer the list builder "
                    "allocated while the original rows were still resident (before dph_index_rehome_rows)")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import Shard
    from densephrases_amd.ivf import assign_lists_resident, train_centroids_resident
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    n = args.rows // 32 * 32
    s = Shard(n, device=0)
    s.fill_synthetic(seed=42, kind=args.kind)
    rng = np.random.default_rng(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_info = None
    if args.centroids == "random":
        cent = rng.normal(0, 0.5, (args.nlist, 768)).astype(np.float32)
    else:
        cent, train_info = train_centroids_resident(s, args.nlist, iters=args.iters, train_rows=args.train_rows or None,
                                                    return_info=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    assign = assign_lists_resident(s, cent)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    s.make_list_major(assign.data_ptr(), cent, stream=st)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    counts = torch.bincount(assign.to(torch.int64), minlength=args.nlist)
    largest, smallest = int(counts.max().item()), int(counts.min().item())
    del counts
    s.finalize()
    R, k = 2 * args.batch, 10
    x = torch.from_numpy(make_queries(args.kind, n, R, 7, args.queries)).to(dev)
    D = torch.empty((R, k), dtype=torch.float32, device=dev)
    I = torch.empty((R, k), dtype=torch.int64, device=dev)
    status = torch.empty(R, dtype=torch.int32, device=dev)
    runs = (("ivf", lambda: s.search_ivf_dev(x.data_ptr(), R, k, args.nprobe, D.data_ptr(), I.data_ptr(), status.data_ptr())),
            ("exact", lambda: s.search_dev(x.data_ptr(), R, k, D.data_ptr(), I.data_ptr(), status.data_ptr())))

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s.profile_enable(True)
        s.profile_read()
        t = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / args.steps
        scan_ms, scan_n = s.profile_read()
        s.profile_enable(False)
        return dt, scan_ms / args.steps, scan_n / args.steps

    before = None
    if args.before_rehome:
        before = {}
        for name, fn in runs:
            dt, scan_ms, _ = timed(fn)
            before[name] = {"ms_per_batch": dt * 1e3, "full_scan_ms_per_batch": scan_ms}
    del assign
    torch.cuda.empty_cache()
    t4 = time.perf_counter()
    s.rehome_rows(stream=st)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    s.finalize()
    out, ids = {}, {}
    for name, fn in runs:
        dt, scan_ms, scan_n = timed(fn)
        wp = s.wave_pairs(0).astype(np.int64)
        wr = s.wave_pairs(1).astype(np.int64)
        out[name] = {"ms_per_batch": dt * 1e3, "queries_per_sec": args.batch / dt, "full_scan_ms_per_batch": scan_ms,
                     "full_scans_per_batch": scan_n, "status_zero_rows": int((status == 0).sum().item()), "stats_last_call": s.stats(),
                     "pairs_last_first_attempt_scan": int(wp.sum()), "max_pairs_in_one_scan_wave": int(wp.max()),
                     "scan_waves_above_8192_pairs": int((wp > 8192).sum()), "pairs_last_retry_scan": int(wr.sum())}
        ids[name] = I.clone()
    units = s.debug_units()
    a, b = ids["ivf"], ids["exact"]
    recall = {}
    for kk in (1, 5, 10):
        hit = (a[:, :kk, None] == b[:, None, :kk]).any(1).float().sum(1) / kk      # share of the exact top-kk found in IVF's top-kk
        recall[f"recall_at_{kk}"] = float(hit.mean().item())
    sweep = None
    if args.sweep:
        xs = torch.from_numpy(make_queries(args.kind, n, R, 11, "near", noise=args.sweep_noise)).to(dev)
        s.search_dev(xs.data_ptr(), R, k, D.data_ptr(), I.data_ptr(), status.data_ptr())
        torch.cuda.synchronize()
        I_ex = I.clone()
        sweep = {"query_noise": args.sweep_noise, "exact_status_zero_rows": int((status == 0).sum().item()), "nprobe": {}}
        for npb in [int(v) for v in args.sweep.split(",")]:
            fn = lambda: s.search_ivf_dev(xs.data_ptr(), R, k, npb, D.data_ptr(), I.data_ptr(), status.data_ptr())      # noqa: E731
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 3
            rec = {}
            for kk in (1, 5, 10):
                hit = (I[:, :kk, None] == I_ex[:, None, :kk]).any(1).float().sum(1) / kk
                rec[f"recall_at_{kk}"] = float(hit.mean().item())
            sweep["nprobe"][str(npb)] = {"ms_per_batch": dt * 1e3, "queries_per_sec": args.batch / dt,
                                         "status_zero_rows": int((status == 0).sum().item()), **rec}
    print(json.dumps({"sweep": sweep, "rows": n, "kind": args.kind, "centroids": args.centroids, "queries": args.queries, "nlist": args.nlist,
                      "train_seconds": t1 - t0, "train": train_info, "assign_seconds": t2 - t1, "list_builder_seconds": t3 - t2,
                      "rehome_seconds": t5 - t4, "largest_list": largest, "smallest_list": smallest, "nprobe": args.nprobe,
                      "batch": args.batch, "before_rehome": before, **out, "unit_queue": units, "ivf_vs_exact": recall}))


if __name__ == "__main__":
    main()
