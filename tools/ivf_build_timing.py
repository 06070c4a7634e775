#!/usr/bin/env python3
"""Full-size IVF build where the rows lie: a synthetic flat shard resident in HBM -> list assignment (fused MFMA GEMM +
arg-max over the int8 rows) -> device-side list builder (radix sort + gather) -> finalize -> one IVF search batch
against the exact search over the same (now list-major) shard.  Prints one JSON line with the phase times."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=170_000_000)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=256)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--cooldown", type=float, default=20.0, help="seconds of idle before the search timings are repeated "
                    "(the assignment runs the f32 MFMA flat out for seconds: tells a clock drop from a real slowdown)")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import Shard
    from densephrases_amd.ivf import assign_lists_resident
    dev = torch.device("cuda", 0)
    n = args.rows // 32 * 32
    s = Shard(n, device=0)
    s.fill_synthetic(seed=42)
    rng = np.random.default_rng(0)
    cent = rng.normal(0, 0.5, (args.nlist, 768)).astype(np.float32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assign = assign_lists_resident(s, cent)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    s.make_list_major(assign.data_ptr(), cent, stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    s.finalize()
    t3 = time.perf_counter()
    counts = torch.bincount(assign.to(torch.int64), minlength=args.nlist)
    R, k = 2 * args.batch, 10
    x = torch.from_numpy(rng.normal(0, 0.5, (R, 768)).astype(np.float32)).to(dev)
    D = torch.empty((R, k), dtype=torch.float32, device=dev)
    I = torch.empty((R, k), dtype=torch.int64, device=dev)
    st = torch.empty(R, dtype=torch.int32, device=dev)
    out = {}
    for name, fn in (("ivf", lambda: s.search_ivf_dev(x.data_ptr(), R, k, args.nprobe, D.data_ptr(), I.data_ptr(), st.data_ptr())),
                     ("exact", lambda: s.search_dev(x.data_ptr(), R, k, D.data_ptr(), I.data_ptr(), st.data_ptr()))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 4
        pairs, triggers = s.scan_counters()
        out[name] = {"ms_per_batch": dt * 1e3, "queries_per_sec": args.batch / dt, "certified": int((st == 0).sum().item()),
                     "stats_last_call": s.stats(), "scan_pairs_last_pass": int(pairs), "top1": I[:, 0].clone()}
    if args.cooldown > 0:
        time.sleep(args.cooldown)
        for name, fn in (("ivf", lambda: s.search_ivf_dev(x.data_ptr(), R, k, args.nprobe, D.data_ptr(), I.data_ptr(), st.data_ptr())),
                         ("exact", lambda: s.search_dev(x.data_ptr(), R, k, D.data_ptr(), I.data_ptr(), st.data_ptr()))):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            out[name]["ms_per_batch_after_cooldown"] = (time.perf_counter() - t) / 4 * 1e3
            out[name]["stats_after_cooldown"] = s.stats()
    recall1 = float((out["ivf"]["top1"] == out["exact"]["top1"]).float().mean().item())
    for v in out.values():
        del v["top1"]
    print(json.dumps({"rows": n, "nlist": args.nlist, "assign_seconds": t1 - t0, "list_builder_seconds": t2 - t1,
                      "finalize_seconds": t3 - t2,
                      "largest_list": int(counts.max().item()), "smallest_list": int(counts.min().item()),
                      "nprobe": args.nprobe, "batch": args.batch, **out, "recall_at_1_vs_exact_random_queries": recall1}))


if __name__ == "__main__":
    main()
