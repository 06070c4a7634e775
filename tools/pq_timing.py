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
    ap.add_argument("--skew", default="", help="list-size skew of the synthetic index: '' (near-uniform), zipf, lognormal, giant (synth._pq_list_sizes); long lists are the most probed (--hot)")
    ap.add_argument("--hot", type=float, default=0.18)
    ap.add_argument("--phases", action="store_true", help="phase clock of the row-major ADC scan (dph_debug_pq_phases) of one more batch")
    ap.add_argument("--tune", action="append", default=[], help="libdph tuning key=v (e.g. coarse_filter=0)")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from densephrases_amd.synth import synthetic_pq_shard
    rng = np.random.default_rng(1)
    n, nlist, M = args.codes, args.nlist, args.M
    t0 = time.perf_counter()
    s, A, cent, sizes = synthetic_pq_shard(n, nlist, M, device=0, skew=args.skew, hot=args.hot)
    torch.cuda.synchronize()
    load_s = time.perf_counter() - t0
    dev = torch.device("cuda", 0)
    for t in args.tune:
        key, _, vals = t.partition("=")
        s.set_tuning(key, *[int(v) for v in vals.split(",") if v != ""])
    s.profile_enable(True)
    srt = np.sort(sizes)[::-1]
    out = {"tune": args.tune, "skew": args.skew or "none", "hot": args.hot if args.skew else None,
           "list_sizes": {"mean": float(sizes.mean()), "max": int(srt[0]), "top5": [int(v) for v in srt[:5]], "lists_over_100x_mean": int((sizes > 100 * sizes.mean()).sum()),
                          "codes_in_256_longest": int(srt[:256].sum())}, "codes": n, "nlist": nlist, "nprobe": args.nprobe, "M": M, "load_seconds": load_s, "batches": {}}
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
        gemm_ms, gemm_n = s.profile_read()
        phases = None
        if args.phases:
            s.debug_pq_phases(0)                     # arms the clocks
            s.debug_pq_phases(1)
            fn()
            torch.cuda.synchronize()
            sel = s.debug_pq_phases(1)[:R]
            stamps = sel[:, :7].astype(np.float64)
            d = np.diff(stamps, axis=1) / 100.0      # us between stamps
            done = stamps[:, 6] > 0
            select_phases = {"rows": int(R), "rows_that_ranked_a_band": int(done.sum()),
                             "us_mean": {k: float(d[done, i].mean()) if done.any() else None for i, k in
                                         enumerate(["query_norm", "candidates_to_lds", "kth_candidate", "marking", "band_dots", "band_ranks"])},
                             "band_lists_mean": float((sel[:, 7] & np.uint64(0xFFFF)).mean()),
                             "candidates_mean": float(((sel[:, 7] >> np.uint64(16)) & np.uint64(0xFFFFFF)).mean()),
                             "needed_from_band_mean": float((sel[:, 7] >> np.uint64(40)).mean())}
            ph = s.debug_pq_phases(0).astype(np.float64)
            ph = ph[ph[:, 6] > 0]
            if len(ph):
                us = lambda c: {"mean": float(c.mean()) / 100.0, "max": float(c.max()) / 100.0}     # noqa: E731  (100 MHz ticks)
                phases = {"workgroups_with_work": int(len(ph)), "busy_us": us(ph[:, 1] - ph[:, 0]),
                          "kernel_span_us": float(ph[:, 1].max() - ph[:, 0].min()) / 100.0,
                          "table_and_lists_us": us(ph[:, 2]), "dis0_us": us(ph[:, 3]), "sums_us": us(ph[:, 4]), "select_append_us": us(ph[:, 5]),
                          "units": us(ph[:, 6] * 100.0), "codes": us(ph[:, 7] * 100.0)}
            s.profile_read()
        failed_over, emitted = s.debug_pq_coarse()
        # codes a batch scores: every query row scans its nprobe lists
        probe = torch.topk((x @ torch.from_numpy(A).to(dev).T) @ torch.from_numpy(cent).to(dev).T, min(args.nprobe, nlist), dim=1).indices
        psz = torch.from_numpy(sizes).to(dev)[probe]
        scanned = float(psz.sum().item())
        per_row = psz.sum(1).double()
        seg = 12288 if M <= 96 else 6144
        n_units = float(torch.ceil(psz.view(R, -1, 64).sum(2).double() / seg).sum().item()) if min(args.nprobe, nlist) % 64 == 0 else None
        gathers = scanned * M
        lds_peak = 256 * 64 * 2.4e9          # CUs x 64 dwords per clock x 2.4 GHz (MI355X_MICROARCH.md LDS section), conflict-free
        out["batches"][str(B)] = {"coarse_filter_gemm_ms": gemm_ms / gemm_n if gemm_n else None, "coarse_failed_over": failed_over, "adc_phases": phases, "select_phases": select_phases if args.phases else None,
                                  "coarse_candidates_per_row": emitted / R if emitted else None, "ms_per_batch": dt * 1e3, "queries_per_sec": B / dt, "status_zero_rows": int((st == 0).sum().item()),
                                  "codes_scored_per_batch": scanned, "codes_per_row": {"mean": float(per_row.mean().item()), "max": float(per_row.max().item())},
                                  "rows_probing_the_longest_list": int((probe == int(np.argmax(sizes))).any(1).sum().item()), "units_of_12288_codes": n_units,
                                  "ns_per_code_per_workgroup": dt * 1e9 * 256 / scanned, "lds_gathers_per_sec": gathers / dt,
                                  "roofline": {"bound": "lds-gather", "achieved": gathers / dt / 1e12, "peak": lds_peak / 1e12,
                                               "unit": "T look-ups/s", "frac": gathers / dt / lds_peak},
                                  "code_bytes_once": float(n) * M, "code_bytes_if_every_row_read_its_lists": scanned * M}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
