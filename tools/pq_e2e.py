#!/usr/bin/env python3
"""The reference's shipping configuration end to end, alone (for a kernel trace): the synthetic OPQ96 / 2^20-list / 170 M-code index
with idx2id + f2o (synth.synthetic_pq_shard, doc_len 100), then bench.pq_e2e -- MIPS.search, MIPS.search_stream and the GPU half alone.
    python tools/pq_e2e.py [--codes N] [--nlist L] > gpurun_out/r05_pq_e2e.json
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pq_e2e -- python tools/pq_e2e.py   (per-kernel microseconds incl. pq_window_kernel)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--codes", type=int, default=170_000_000)
    ap.add_argument("--nlist", type=int, default=1 << 20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--top_k", type=int, default=10)
    ap.add_argument("--max_answer_length", type=int, default=10)
    ap.add_argument("--b512", action="store_true", help="also bench.pq_b512_document: configs[4] over the index (batch 512, document units, stream) with the host half's breakdown")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from densephrases_amd.synth import synthetic_pq_shard
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    s, A, cent, sizes = synthetic_pq_shard(a.codes, a.nlist, 96, device=0, doc_len=100)
    torch.cuda.synchronize()
    load_s = time.perf_counter() - t0
    R, k = 2 * a.batch, a.top_k
    x = torch.from_numpy(np.random.default_rng(3).normal(0, 0.5, (R, 768)).astype(np.float32)).to(dev)
    D = torch.empty((R, k), dtype=torch.float32, device=dev)
    I = torch.empty((R, k), dtype=torch.int64, device=dev)
    st = torch.empty(R, dtype=torch.int32, device=dev)
    for _ in range(3):
        s.search_ivf_dev(x.data_ptr(), R, k, 256, D.data_ptr(), I.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        s.search_ivf_dev(x.data_ptr(), R, k, 256, D.data_ptr(), I.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    a.no_check = False
    out = bench.pq_e2e(s, a, dev, a.batch / dt)
    out["search_only_queries_per_sec"] = a.batch / dt
    out["index_load_seconds"] = load_s
    if a.b512:
        out["b512_document_stream"] = [bench.pq_b512_document(s, a, dev) for _ in range(2)]        # (twice: the host half varies run to run)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
