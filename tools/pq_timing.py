#!/usr/bin/env python3
"""Throughput of the IVFPQ search (csrc/dph_pq.hip: the reference's own index type, OPQ96 + IVF + 8-bit PQ, resident in HBM)
at full size.  Timing only: a synthetic index of --codes random PQ codes in --nlist lists (random lengths), random
codebooks / coarse centroids, a Householder rotation as the OPQ matrix -- what the ADC scan costs does not depend on
what the codes mean (parity: tests/test_pq.py on trained indexes).  Prints one JSON line: ms per batch, queries/sec, the
ADC kernel's share, LDS gathers per second against the LDS peak (the scan is LDS-gather-bound: M look-ups per code) and the
HBM bytes the codes amount to."""
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
    ap.add_argument("--codes", type=int, default=170_000_000)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=256)
    ap.add_argument("--M", type=int, default=96)
    ap.add_argument("--batches", default="64,256")
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import Shard
    from oracle.make_golden_pq import householder_rotation
    rng = np.random.default_rng(0)
    n, nlist, M = args.codes, args.nlist, args.M
    # list lengths: exponential weights (a few long lists, many short ones), summing to n
    w = rng.exponential(1.0, nlist)
    sizes = np.floor(w / w.sum() * n).astype(np.int64)
    sizes[0] += n - int(sizes.sum())
    A = householder_rotation(rng.normal(0, 1, 768).astype(np.float32), rng.permutation(768))
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    pqc = rng.normal(0, 0.1, (M, 256, 768 // M)).astype(np.float32)
    # the codes: one random block, rolled per chunk (generation at ~1 GB/s would dominate the run otherwise)
    block = rng.integers(0, 256, (1 << 20, M), dtype=np.uint8)

    # build the shard by hand: the lists are generated one at a time, never whole in host memory
    import ctypes as C
    from densephrases_amd import _lib
    t0 = time.perf_counter()
    s = Shard.__new__(Shard)
    s._h = C.c_void_p()
    _lib._chk(_lib.lib.dph_index_create_pq(0, n, nlist, M, C.byref(s._h)))
    s.device, s.id_base, s.n_rows = 0, 0, n
    _lib._chk(_lib.lib.dph_index_set_pq(s._h, _lib._p(np.ascontiguousarray(A)), None, _lib._p(cent), _lib._p(np.ascontiguousarray(pqc)), 1))
    _lib._chk(_lib.lib.dph_index_set_pq_list_sizes(s._h, _lib._p(sizes)))
    # positions are list-major and contiguous: upload in chunks of the random block, whatever list they fall into
    for pos in range(0, n, block.shape[0]):
        m = min(block.shape[0], n - pos)
        c = np.ascontiguousarray(np.roll(block, pos // block.shape[0] % 97, axis=0)[:m])
        ids = np.arange(pos, pos + m, dtype=np.int64)
        _lib._chk(_lib.lib.dph_index_upload_pq_codes(s._h, pos, m, _lib._p(c), _lib._p(ids)))
    s.pq = {"nlist": nlist, "M": M, "nprobe": 256}
    s.set_idx2id(np.zeros(n, np.int32), np.zeros(n, np.int32))
    s.set_f2o(np.zeros(1, np.int32), np.asarray([0, 1], np.int64), np.zeros(1, np.int32))
    s.finalize()
    torch.cuda.synchronize()
    load_s = time.perf_counter() - t0
    dev = torch.device("cuda", 0)
    out = {"codes": n, "nlist": nlist, "nprobe": args.nprobe, "M": M, "load_seconds": load_s, "batches": {}}
    k = 10
    for B in [int(b) for b in args.batches.split(",")]:
        R = 2 * B
        x = torch.from_numpy(rng.normal(0, 0.5, (R, 768)).astype(np.float32)).to(dev)
        D = torch.empty((R, k), dtype=torch.float32, device=dev)
        I = torch.empty((R, k), dtype=torch.int64, device=dev)
        st = torch.empty(R, dtype=torch.int32, device=dev)
        fn = lambda: s.search_ivf_dev(x.data_ptr(), R, k, args.nprobe, D.data_ptr(), I.data_ptr(), st.data_ptr())     # noqa: E731
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / args.steps
        # codes a batch scores: every query row scans its nprobe lists
        probe = torch.topk((x @ torch.from_numpy(A).to(dev).T) @ torch.from_numpy(cent).to(dev).T, min(args.nprobe, nlist), dim=1).indices
        scanned = float(torch.from_numpy(sizes).to(dev)[probe.flatten()].sum().item())
        gathers = scanned * M
        lds_peak = 256 * 64 * 2.4e9          # CUs x 64 dwords per clock x 2.4 GHz (MI355X_MICROARCH.md LDS section), conflict-free
        out["batches"][str(B)] = {"ms_per_batch": dt * 1e3, "queries_per_sec": B / dt, "status_zero_rows": int((st == 0).sum().item()),
                                  "codes_scored_per_batch": scanned, "lds_gathers_per_sec": gathers / dt,
                                  "roofline": {"bound": "lds-gather", "achieved": gathers / dt / 1e12, "peak": lds_peak / 1e12,
                                               "unit": "T look-ups/s", "frac": gathers / dt / lds_peak},
                                  "code_bytes_once": float(n) * M, "code_bytes_if_every_row_read_its_lists": scanned * M}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
