#!/usr/bin/env python3
"""Diagnosis of the full-size list-major slowdown (a batch of 256 through IVF takes 63 ms on a shard
BUILT in HBM against 23 ms on random contiguous lists over the original buffer, whatever the data and the queries).
Stages, each timing the same two searches (IVF-4096 / nprobe 256 and exact, batch 256) and a plain torch streaming read
of the shard's row buffer:
  flat      the freshly filled flat shard (exact only)
  built     after dph_index_make_list_major with a RANDOM assignment (no k-means, no assignment GEMM before it: the chip
            has done nothing heavy), on the buffer the builder allocated next to the original
  rehomed   after dph_index_rehome_rows
  rested    after --rest seconds of idling
rocm-smi clocks / power are sampled once per second into gpurun_out/ by the caller (tools/r03_trip4.sh)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Dev64:
    def __init__(self, ptr, n_bytes):
        self.__cuda_array_interface__ = {"shape": (n_bytes // 8,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=170_000_000)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--rest", type=float, default=10.0)
    ap.add_argument("--assign", default="random", choices=["random", "gemm"])
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import Shard
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    n = args.rows // 32 * 32
    s = Shard(n, device=0)
    s.fill_synthetic(seed=42, kind=0)
    s.finalize()
    rng = np.random.default_rng(0)
    R, k = 512, 10
    x = torch.from_numpy(rng.normal(0, 0.5, (R, 768)).astype(np.float32)).to(dev)
    D = torch.empty((R, k), dtype=torch.float32, device=dev)
    I = torch.empty((R, k), dtype=torch.int64, device=dev)
    status = torch.empty(R, dtype=torch.int32, device=dev)
    ivf = lambda: s.search_ivf_dev(x.data_ptr(), R, k, 256, D.data_ptr(), I.data_ptr(), status.data_ptr())      # noqa: E731
    exact = lambda: s.search_dev(x.data_ptr(), R, k, D.data_ptr(), I.data_ptr(), status.data_ptr())           # noqa: E731

    def timed(fn, steps=3):
        fn()
        torch.cuda.synchronize()
        s.profile_enable(True)
        s.profile_read()
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        ms, cnt = s.profile_read()
        s.profile_enable(False)
        return {"ms_per_batch": dt * 1e3, "scan_ms_per_launch": ms / max(cnt, 1), "fast": s.stats()["certified_fast"]}

    def stream_read():
        ptr = s.rows_dev_ptr()
        t = torch.as_tensor(_Dev64(ptr, int(s.n_rows) * 768), device=dev)
        t[: 1 << 27].sum()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t.sum()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s.finalize()
        return {"torch_sum_GBps": t.numel() * 8 / dt / 1e9}

    out = {"rows": n, "assign": args.assign}
    out["flat"] = {"exact": timed(exact), **stream_read()}
    cent = rng.normal(0, 0.5, (args.nlist, 768)).astype(np.float32)
    if args.assign == "random":
        # lists of very different lengths like real ones: list = floor(nlist * u^2)
        u = torch.rand(n, device=dev)
        assign = (u * u * args.nlist).to(torch.int32).clamp_(max=args.nlist - 1)
        del u
    else:
        from densephrases_amd.ivf import assign_lists_resident
        assign = assign_lists_resident(s, cent)
    torch.cuda.synchronize()
    out["free_gib_before_build"] = torch.cuda.mem_get_info(dev)[0] / 2**30
    t0 = time.perf_counter()
    s.make_list_major(assign.data_ptr(), cent, stream=st)
    torch.cuda.synchronize()
    out["list_builder_seconds"] = time.perf_counter() - t0
    del assign
    torch.cuda.empty_cache()
    s.finalize()
    out["free_gib_after_build"] = torch.cuda.mem_get_info(dev)[0] / 2**30
    out["built"] = {"ivf": timed(ivf), "exact": timed(exact), **stream_read()}
    s.rehome_rows(stream=st)
    torch.cuda.synchronize()
    s.finalize()
    out["rehomed"] = {"ivf": timed(ivf), "exact": timed(exact), **stream_read()}
    time.sleep(args.rest)
    out["rested"] = {"ivf": timed(ivf), "exact": timed(exact)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
