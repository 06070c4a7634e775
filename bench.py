#!/usr/bin/env python3
"""bench.py -- queries/sec of the DensePhrases phrase-retrieval hot path (MIPS.search: dense top-k over the int8
phrase dump + start/end window re-scoring) on MI355X, one process per GPU.

A *step* = one batch of B queries ([2B,768] stacked start/end rows, reference index.py:196-200) through
  dph_search_dev  (quantise -> sampled bounds -> int8 MFMA filter scan of the whole resident shard -> refine ->
                   select / exact fp64 re-rank / certify -> on-device retry of uncertified rows)
  dph_rescore_dev x2 (window re-score of the 2*B*k candidates, L = 10)
  [N > 1]  two small RCCL all-gathers (sample scores for the union bound, then every rank's [2B,k] record) + merge.
Inputs are resident in HBM before the timed region.  Workload at N=1 = BASELINE.json configs[1]: 1 x MI355X,
brute-force IP, batch 64, the 170 M-row int8 dump (synthetic, generated on-device).  For N > 1 the default is
BASELINE.json configs[2]'s sizing: 162.5 M rows PER GPU (1.3 B rows over 8 GPUs), range-partitioned, `"scaling": "weak"`
(per-GPU work fixed; every query searches every shard, so the ideal is a CONSTANT queries/sec while the dump grows N-fold
-- `row_queries_per_sec` = rows_total x queries/sec is the aggregate that grows with N).  An explicit `--rows R` partitions
R rows over the ranks instead (strong scaling: rows/GPU = R / N).  `--batch 256 / 512` are the batch shapes of
configs[3] / configs[4] (passes of 256 query rows: the dump is read once per 256 rows); `--dist mixture` swaps the
i.i.d. dump for the mixture-of-4096-Gaussians + saturated-outlier dump (SURVEY.md 8d, config 4 data).

After the configs[1] measurement (which alone is `value`), an N=1 run times three more workloads and appends them under
`also` (skip with --no_also): `pq_opq96_ivf2p20_b64` (the reference's own index type -- OPQ96 + IVFPQ, 2^20 lists, nprobe 256 --
resident in HBM), `e2e_mips_search` (host queries in, result dicts out through the python class the reference's
callers use), `exact_b512_document` (configs[4]'s shape: batch 512, retrieval_unit=document => top_k doubled, title
de-duplication, through MIPS.search_stream) and `ivf4096_b256` (configs[3]: the mixture dump, k-means lists BUILT in
HBM, nprobe 256, batch 256, roofline on the PROBED bytes, recall against the exact search of the same run), `anisotropic_b64`
(configs[1]'s shape over the BERT-like dump: rogue dimensions, log-normal row norms, near-duplicate runs -- the filter's stress test)
and, inside the PQ leg, `e2e_mips_search` over the PQ index (the reference's shipping configuration end to end).

Invoked as `python bench.py --gpus N` with N > 1 and no torchrun environment, it spawns the N ranks itself.

Prints ONE JSON line on rank 0 (see the contract in the task statement) with `roofline` (the scan kernel: algorithmic
bytes per launch / average launch duration from HIP events recorded around every full-scan launch on its stream) and
`cpu_baseline` (a FAISS-CPU-shaped fp32 flat search on a bounded sample, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
I8_MFMA_PEAK_TOPS = 5000.0     # dense int8 MFMA = 2x the 2.5 PFLOP/s bf16 figure of the guide (measured floor there: 4404)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=15)       # 1000 NQ questions / 64 ~ 15 full batches
    ap.add_argument("--warmup", type=int, default=5)       # run_demo.py:332-352 excludes the first 5 batches
    ap.add_argument("--rows", type=int, default=0, help="total dump rows (all ranks together); default: 170 M at N=1 "
                    "(configs[1]), 162.5 M per GPU at N>1 (configs[2], weak scaling)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--top_k", type=int, default=10)
    ap.add_argument("--max_answer_length", type=int, default=10)
    ap.add_argument("--dist", choices=["iid", "mixture", "docruns", "anisotropic"], default="iid",
                    help="synthetic dump: i.i.d., mixture of 4096 Gaussians + saturated outliers, document-ordered runs of near-duplicates, or the "
                         "BERT-like dump (rogue dimensions, log-normal row norms, near-duplicate runs; synth.py kind 4)")
    ap.add_argument("--cpu_gib", type=float, default=8.0, help="fp32 GiB of the bounded CPU-baseline sample")
    ap.add_argument("--no_also", action="store_true", help="skip the configs[3]/[4]/end-to-end sub-records")
    ap.add_argument("--no_traffic", action="store_true", help="skip the nested rocprofv3 FETCH_SIZE pass behind roofline.traffic")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no_check", action="store_true", help="skip the result assertions (timing experiments only)")
    ap.add_argument("--tune", action="append", default=[], help="libdph tuning key=v[,v..] (dph_index_set_tuning)")
    ap.add_argument("--per_step", action="store_true", help="diagnostic: synchronise after every step and print its wall time to stderr")
    ap.add_argument("--queries", choices=["planted", "encoder"], default="planted",
                    help="query batches: half planted near stored rows + half random (default), or encoder-like (docruns / anisotropic dumps: "
                         "every query the noisy mean of a near-duplicate run, both halves planted)")
    ap.add_argument("--trace_out", default="", help="write PATH: per timed full-scan launch the HIP-event duration and the rocprofv3 --kernel-trace dispatch "
                    "duration of the same workload on this box (tools/trace_out.py; `--dist anisotropic`: that leg), then exit")
    ap.add_argument("--recall_queries", type=int, default=64,
                    help="queries of the last batch whose top-k is recomputed by an independent fp64 scan for recall@k")
    return ap.parse_args()


def cpu_baseline(args, n_total):
    """The FAISS-CPU IndexFlatIP execution shape on a bounded sample, all host cores, in processes of their own -- THREE implementations,
    the fastest is `value`, the others `alt`:
      c_avx512  oracle/cpu_baseline_c.py: plain C + OpenMP with an AVX-512 sgemm micro-kernel (oracle/csrc/cpu_flat_avx512.c) -- what a
             FAISS build with a good BLAS would do on this host;
      torch  oracle/cpu_baseline_torch.py: what BASELINE.md section 3 prescribes -- `torch.mm` + `torch.topk` (MKL / oneDNN sgemm,
             torch.set_num_threads(all cores)), database blocks of 1024 / 8192 / 65536 rows, the best block size;
      numpy  oracle/cpu_baseline.py: one single-threaded OpenBLAS sgemm per 8192-row block on one python thread per core (FAISS'
             OpenMP-over-blocks shape), running top-k per thread, merged at the end."""
    def run(mod):
        # (a comparator that hangs or fails must never take the bench line with it: bounded, every failure becomes an "error" entry)
        try:
            r = subprocess.run([sys.executable, "-m", mod, "--batch", str(args.batch), "--top_k", str(args.top_k),
                                "--gib", str(args.cpu_gib), "--budget", "12"], cwd=ROOT, capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return {"error": (r.stderr or r.stdout)[-400:]}
            return json.loads(r.stdout.strip().splitlines()[-1])
        except (subprocess.TimeoutExpired, OSError, ValueError, IndexError) as e:
            return {"error": repr(e)[:300]}

    res = {"c_avx512": run("oracle.cpu_baseline_c"), "torch": run("oracle.cpu_baseline_torch"), "numpy": run("oracle.cpu_baseline")}
    ok = {k: v for k, v in res.items() if "error" not in v}
    if not ok:
        return {"value": None, "unit": "queries/sec", "cores": 0, "kind": "port", "sample": "both CPU comparators failed", "errors": res}
    best = max(ok, key=lambda k: ok[k]["qps_sample"])
    m = ok[best]

    def describe(name, v):
        if "error" in v:
            return f"{name}: failed ({v['error'][-120:]})"
        per_core = v["gflops"] / v["cores"]
        how = ("torch.mm + torch.topk, MKL sgemm over all threads, best of blocks " + "/".join(sorted(v.get("per_block", {}), key=int)) if name == "torch"
               else "plain C, AVX-512 micro-kernel (3 rows x 128 queries in registers), one OpenMP thread per hardware thread or per core, whichever is faster"
               if name == "c_avx512" else "numpy: one single-threaded OpenBLAS sgemm per 8192-row block on one python thread per core")
        why = ""
        if v.get("cpu_quota"):
            why = f" (the pod's cgroup CPU quota is {v['cpu_quota']} CPUs of the host's {v['host_cores']} hardware threads: that many threads are used)"
        if per_core < 10.0:
            why += (f" -- below 10 GFLOP/s per thread: a [{v['block']},768]x[768,{2 * args.batch}] product has {2 * args.batch} columns, too thin for "
                   f"{v['cores']} threads to share (and the box has {v['host_cores']} hardware threads on fewer physical cores)")
        return (f"{name} ({how}): {v['qps_sample']:.2f} Q/s on {v['rows']} rows = {v['gflops']:.0f} GFLOP/s = {per_core:.1f} GFLOP/s per thread "
                f"x {v['cores']} threads, {v['db_gbytes_per_s']:.0f} GB/s of fp32 database bytes, block {v['block']}, median of {v['passes']} passes{why}")

    out = {
        "value": m["qps_sample"] * m["rows"] / n_total, "unit": "queries/sec", "cores": m["cores"], "kind": "port", "implementation": best,
        "gflops": m["gflops"], "gflops_per_core": m["gflops"] / m["cores"], "db_gbytes_per_s": m["db_gbytes_per_s"], "host_cores": m["host_cores"],
        "cpu_quota": m.get("cpu_quota"),
        "sample": (f"FAISS-CPU IndexFlatIP's execution shape (fp32 index resident in RAM, {m['sample_gib']:.1f} GiB = {m['rows']} distinct rows of the "
                   f"dump's distribution, past every cache; blocked sgemm + running top-k), own process, B={args.batch}; " + describe(best, m) +
                   f"; value = that rate scaled linearly in N to {n_total} rows.  The other implementation -- " +
                   "; ".join(describe(k, v) for k, v in res.items() if k != best)),
        "alt": {k: ({"value": v["qps_sample"] * v["rows"] / n_total, "gflops": v["gflops"], "gflops_per_core": v["gflops"] / v["cores"], "cores": v["cores"],
                     "block": v["block"], "per_block": v.get("per_block")} if "error" not in v else v) for k, v in res.items() if k != best},
    }
    if "per_block" in m:
        out["per_block"] = m["per_block"]
    return out


def also_e2e(shard, args, n_total):
    """end-to-end MIPS.search as the reference's callers use it (eval_phrase_retrieval.py:72-77) --
    host float queries in, List[List[dict]] out (PCIe copies, idx2id, metadata, dict assembly, paragraph cropping,
    de-duplication) -- over the SAME resident configs[1] shard; run_demo.py:329-352: first 5 batches excluded."""
    from densephrases_amd import MIPS
    from densephrases_amd.synth import SynthDocStore, synthetic_rows
    B, k = args.batch, args.top_k
    mips = MIPS.from_shard(shard, SynthDocStore())
    rng = np.random.default_rng(5)
    batches = []
    for _ in range(4):
        q = rng.normal(0, 0.5, (B, 1536)).astype(np.float32)
        p = rng.integers(0, n_total - 8, B)
        rows = np.stack([synthetic_rows(int(r), 1, args.seed)[0] for r in p]).astype(np.float32) / 20 - 2
        rows_e = np.stack([synthetic_rows(int(r) + 2, 1, args.seed)[0] for r in p]).astype(np.float32) / 20 - 2
        q[:, :768] = rows + rng.normal(0, 0.1, rows.shape)
        q[:, 768:] = rows_e + rng.normal(0, 0.1, rows.shape)
        batches.append((q.astype(np.float64), p))
    steps, warm = 10, 5
    for i in range(warm):
        mips.search(batches[i % 4][0], q_texts=["q"] * B, top_k=k, aggregate=True, agg_strat="opt1")
    t0 = time.perf_counter()
    for i in range(steps):
        out = mips.search(batches[i % 4][0], q_texts=["q"] * B, top_k=k, aggregate=True, agg_strat="opt1")
    dt = time.perf_counter() - t0
    for _ in mips.search_stream((batches[i % 4][0] for i in range(3)), top_k=k, aggregate=True):
        pass
    t0 = time.perf_counter()
    n_out = 0
    for outs in mips.search_stream((batches[i % 4][0] for i in range(steps)), q_texts=(["q"] * B for _ in range(steps)),
                                   top_k=k, aggregate=True, agg_strat="opt1"):
        n_out += len(outs)
    dt_s = time.perf_counter() - t0
    p = batches[(steps - 1) % 4][1]
    ok = sum(1 for r, pr in zip(out, p) if r and r[0]["doc_idx"] == pr // 100 and r[0]["start_idx"] == pr % 100
             and r[0]["end_idx"] == pr % 100 + 2)
    assert args.no_check or (n_out == steps * B and ok >= B * 9 // 10), (n_out, ok)       # (a planted end 2 rows on may cross its document)
    return {"workload": f"MIPS.search end to end over the configs[1] shard: host queries in, result dicts out, batch {B}",
            "queries_per_sec": steps * B / dt, "ms_per_batch": dt / steps * 1e3,
            "search_stream_queries_per_sec": steps * B / dt_s, "search_stream_ms_per_batch": dt_s / steps * 1e3,
            "steps": steps, "warmup": warm, "top1_is_planted_phrase": f"{ok}/{B}"}


def also_b512_document(shard, args, n_total):
    """configs[4]'s shape on one GPU: batch 512 streaming queries, retrieval_unit='document' => search_top_k = 2 * top_k
    (model.py:79-81), title de-duplication (agg_strat opt3), through MIPS.search_stream (GPU half of batch t+1 overlaps
    the host half of batch t).  Exact search: 1024 query rows = 4 passes of 256 rows over the resident dump.
    16 timed batches (the un-overlapped tail -- the host half of the LAST batch -- is charged to all of them), 4 distinct
    batches cycled, the metadata of their 80 k documents fetched before the timed region (what a document costs the first
    time is the store's cost -- here a synthetic generator -- not the path's)."""
    from densephrases_amd import MIPS
    from densephrases_amd.synth import SynthDocStore, synthetic_rows
    B, k = 512, 2 * args.top_k
    mips = MIPS.from_shard(shard, SynthDocStore())
    rng = np.random.default_rng(11)
    batches = []
    for _ in range(4):
        q = rng.normal(0, 0.5, (B, 1536)).astype(np.float32)
        p = rng.integers(0, n_total - 8, B)
        rows = np.stack([synthetic_rows(int(r), 1, args.seed)[0] for r in p]).astype(np.float32) / 20 - 2
        q[:, :768] = rows + rng.normal(0, 0.1, rows.shape)
        batches.append((q, p))
    steps = 16
    for _ in mips.search_stream((batches[i % 4][0] for i in range(5)), top_k=k, aggregate=True, agg_strat="opt3"):
        pass
    shard.profile_read()
    tm = mips.reset_timing()
    fetched0 = mips._host.fetched_docs()
    t0 = time.perf_counter()
    outs_all = []
    for outs in mips.search_stream((batches[i % 4][0] for i in range(steps)), q_texts=(["q"] * B for _ in range(steps)),
                                   top_k=k, aggregate=True, agg_strat="opt3"):
        outs_all.append(outs)
    dt = time.perf_counter() - t0
    scan_ms, scan_n = shard.profile_read()
    p = batches[(steps - 1) % 4][1]
    ok = sum(1 for r, pr in zip(outs_all[-1], p) if r and r[0]["doc_idx"] == pr // 100)
    assert args.no_check or ok >= B - 2, ok
    n_rows_q = 2 * B
    alg_batch = n_total * 768 + n_rows_q * 768 * 4 + n_rows_q * k * 12
    ms = dt / steps * 1e3
    host_ms, wait_ms, enq_ms = (tm[key] / steps * 1e3 for key in ("host_s", "wait_s", "enqueue_s"))
    mfma = 2.0 * n_rows_q * 768 * n_total / (scan_ms / steps / 1e3) / 1e12
    return {"workload": "configs[4] shape on 1 GPU: batch 512 (1024 query rows), retrieval_unit=document (top_k doubled to "
                        f"{k}, agg_strat opt3), MIPS.search_stream, exact search over the configs[1] dump",
            "queries_per_sec": steps * B / dt, "ms_per_batch": ms, "steps": steps,
            "scan_launches_per_batch": scan_n / steps, "scan_ms_per_batch": scan_ms / steps,
            # where a batch's wall time goes on the host thread: enqueueing the GPU half of batch t+1, the host half of batch t
            # (C++ assemble + aggregate of 2*B*k candidates -> python dicts), waiting for the GPU; exposed = what the scans do not hide
            "host_ms_per_batch": host_ms, "enqueue_ms_per_batch": enq_ms, "gpu_wait_ms_per_batch": wait_ms,
            "exposed_host_ms": max(0.0, ms - scan_ms / steps),
            "doc_meta_fetches_in_timed_region": int(mips._host.fetched_docs() - fetched0),
            "roofline": {"bound": "mfma", "kernel": "dph_scan_kernel<2, 4, false, 0, 1, false>", "achieved": mfma, "peak": I8_MFMA_PEAK_TOPS,
                         "unit": "TOP/s", "frac": mfma / I8_MFMA_PEAK_TOPS,
                         "hbm_per_batch_frac": alg_batch / (scan_ms / steps / 1e3) / 1e9 / HBM_PEAK_GBS},
            "top1_doc_is_planted": f"{ok}/{B}"}


def also_encoder_overlap(shard, args, dev):
    """configs[4] with the ENCODER in the loop (reference order: eval_phrase_retrieval.py:71-87 encodes a batch,
    then searches it, open_utils.py:83-101): two random-init BERT-base forwards per batch of 512 x 64 tokens (bf16) on a side stream
    against MIPS.search_stream over the configs[1] shard -- encoder alone, search alone, one after the other, overlapped."""
    from densephrases_amd import MIPS
    from densephrases_amd.encoder_stream import measure_overlap
    from densephrases_amd.synth import SynthDocStore
    mips = MIPS.from_shard(shard, SynthDocStore())
    out = measure_overlap(mips, dev, B=512, T=64, k=2 * args.top_k, steps=12)
    out["workload"] = (f"configs[4] shape on 1 GPU with the query encoder in the loop: {out['encoder']} -> MIPS.search_stream "
                       f"(top_k {2 * args.top_k}, opt3) over the configs[1] dump")
    out["queries_per_sec"] = out["queries_per_sec_overlapped"]
    out["ms_per_batch"] = out["overlapped_ms"]
    return out


def make_batches(args, B, n_total, kind, n, dev, seed=1234, style="planted"):
    """n distinct query batches [B, 1536] on the device + the planted rows of each: synthetic NQ-shaped batches, half of them planted
    near stored rows so the result is checkable; on the anisotropic dump the other half are random directions that carry the dump's
    rogue dimensions, as vectors of the same encoder do.
    style "encoder" (document-ordered dumps, kinds 2 / 4): EVERY query is shaped like an encoder's output for a question about a stored
    passage -- the start half is the mean of up to 8 consecutive rows of one near-duplicate run (de-quantised) plus N(0, 0.05^2), the end
    half the same for the rows two further on: both halves land in the middle of a dense neighbourhood (a whole run scores within a few
    per cent of the best row), which is what moves pairs per launch and first-attempt certificates (run_demo.py:329-352 times real
    questions; there are none offline)."""
    import torch
    from densephrases_amd.synth import ROGUE_DIMS, ROGUE_MEANS, synthetic_rows, synthetic_run_of_row
    rng = np.random.default_rng(seed)
    batches, planted = [], []
    for _ in range(n if style == "encoder" else 0):
        q = np.empty((B, 1536), np.float32)
        p = rng.integers(0, n_total - 16, B)
        for i, r in enumerate(p):
            rows = synthetic_rows(int(r), 12, args.seed, kind).astype(np.float32) / 20 - 2
            run = synthetic_run_of_row(np.arange(int(r), int(r) + 12), args.seed)
            same = run == run[0]
            a = rows[:8][same[:8]]
            b = rows[2:10][same[2:10]] if same[2:10].any() else a
            q[i, :768] = a.mean(0) + rng.normal(0, 0.05, 768)
            q[i, 768:] = b.mean(0) + rng.normal(0, 0.05, 768)
        batches.append(torch.from_numpy(q).to(dev))
        planted.append(p)
    for _ in range(0 if style == "encoder" else n):                          # (cycled by the callers; all resident before timing)
        q = rng.normal(0, 0.5, (B, 1536)).astype(np.float32)
        p = rng.integers(0, n_total, B // 2)
        rows = np.stack([synthetic_rows(int(r), 1, args.seed, kind)[0] for r in p]).astype(np.float32) / 20 - 2
        q[:B // 2, :768] = rows + rng.normal(0, 0.1, rows.shape).astype(np.float32)
        if kind == 4:
            rm = (np.asarray(ROGUE_MEANS, np.float32) - 40.0) / 20.0
            for half in (0, 768):
                q[B // 2:, [half + d for d in ROGUE_DIMS]] = rm[None, :] * (1.0 + rng.normal(0, 0.2, (B - B // 2, len(rm)))).astype(np.float32)
            q[:B // 2, 768:][:, list(ROGUE_DIMS)] = rm[None, :]
        batches.append(torch.from_numpy(q).to(dev))
        planted.append(p)
    return batches, planted


def also_anisotropic(args, dev, local):
    """VERDICT r4 item 1 in the driver's own run: configs[1]'s shape (170 M rows, batch 64, k 10, L 10) over the BERT-LIKE dump
    (synth.py kind 4: five rogue dimensions whose code sits near +-100 for every row, log-normal row norms, runs of near-duplicates;
    queries = stored row + noise and rogue-dimension-heavy random directions).  Same step as the headline (`--dist anisotropic` is
    this leg as a line of its own): first-attempt certificates over ALL timed steps, pairs per launch, the aux layout libdph chose,
    recall against an independent fp64 brute force."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.dist import ShardedSearcher
    n, B, k, L, kind = 170_000_000, args.batch, args.top_k, args.max_answer_length, 4
    s = Shard(n, device=local)
    s.fill_synthetic(seed=args.seed, kind=kind)
    s.set_idx2id(((np.arange(n, dtype=np.int64)) // 100).astype(np.int32), ((np.arange(n, dtype=np.int64)) % 100).astype(np.int32))
    nd = n // 100
    s.set_f2o(np.arange(nd + 1, dtype=np.int32), np.arange(0, (nd + 2) * 100, 100, dtype=np.int64), np.tile(np.arange(100, dtype=np.int32), nd + 1))
    t0 = time.perf_counter()
    s.finalize()
    fin_s = time.perf_counter() - t0
    lay = s.aux_layout()
    ss = ShardedSearcher(s, B, k, L, device=dev)
    batches, _ = make_batches(args, B, n, kind, 4, dev)
    s.profile_enable(True)
    for i in range(4):
        out = ss.step(batches[i])
    torch.cuda.synchronize()
    s.profile_read()
    steps = 12
    fast, n_fail = 0, torch.zeros((), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    for i in range(steps):
        out = ss.step(batches[i % 4])
        n_fail += (out["status"] != 0).sum()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    scan_ms, scan_n = s.profile_read()
    st = s.stats()
    pairs, triggers = s.scan_counters()
    # all timed steps certified by the FIRST attempt? (re-run the four batches, reading the statistics of each)
    for i in range(4):
        ss.step(batches[i])
        torch.cuda.synchronize()
        fast += s.stats()["certified_fast"]
    enc, e_got, e_x = None, None, None
    # ---- encoder-like queries over the same shard: every query the noisy mean of a near-duplicate run, both halves
    #      planted -- the whole run scores within a few per cent of the best row: pairs per launch and first-attempt certificates
    try:
        eb, _ = make_batches(args, B, n, kind, 8, dev, seed=4321, style="encoder")
        for i in range(2):
            ss.step(eb[i])
        torch.cuda.synchronize()
        e_fail, e_fast, e_pairs = torch.zeros((), dtype=torch.int64, device=dev), 0, []
        t0 = time.perf_counter()
        for i in range(8):
            o = ss.step(eb[i])
            e_fail += (o["status"] != 0).sum()
        torch.cuda.synchronize()
        e_dt = (time.perf_counter() - t0) / 8
        for i in range(8):                      # (again, one at a time: the statistics of each)
            o = ss.step(eb[i])
            torch.cuda.synchronize()
            e_fast += s.stats()["certified_fast"]
            e_pairs.append(s.scan_counters()[0])
        e_sel = torch.arange(0, 2 * B, 4, device=dev)
        e_got, e_x = o["I"][e_sel].clone(), ss.x[e_sel].clone()
        enc = {"queries": "mean of <= 8 rows of one near-duplicate run + N(0, 0.05^2), start and end half both planted (end = the rows two further on)",
               "queries_per_sec": B / e_dt, "ms_per_batch": e_dt * 1e3, "batches": 8, "uncertified_rows": int(e_fail.item()),
               "certified_by_first_attempt": f"{e_fast}/{8 * 2 * B}", "scan_pairs_per_launch": {"mean": float(np.mean(e_pairs)), "max": int(max(e_pairs))},
               "recall_rows_checked": int(e_sel.numel())}
        assert args.no_check or enc["uncertified_rows"] == 0, enc
    except AssertionError:
        raise
    except Exception as e:                       # the leg's main numbers must survive
        enc = {"error": repr(e)[:300]}
    sel = torch.arange(0, 2 * B, 4, device=dev)
    out = ss.step(batches[3])
    ref_s, ref_i = independent_topk(s.rows_dev_ptr(), n, 0, ss.x[sel], k, dev)
    got = out["I"][sel]
    rec = float((got == ref_i).all(1).float().mean().item())
    rec10 = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(got.cpu().numpy(), ref_i.cpu().numpy())]))
    n_uncert = int(n_fail.item())
    assert args.no_check or (n_uncert == 0 and rec10 == 1.0), (n_uncert, rec10)
    if e_x is not None:                          # (the brute force reads the resident rows: after the last search of the leg)
        _, e_ref_i = independent_topk(s.rows_dev_ptr(), n, 0, e_x, k, dev)
        enc["recall_at_10_vs_fp64_brute_force"] = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(e_got.cpu().numpy(), e_ref_i.cpu().numpy())]))
        assert args.no_check or enc["recall_at_10_vs_fp64_brute_force"] == 1.0, enc
    avg = scan_ms / max(scan_n, 1)
    tiles = (n + 31) // 32
    fused = int(st.get("fused_stride", 0) or 0)
    launch_rows = (tiles - (tiles + fused - 1) // fused) * 32 if fused >= 2 else n
    alg = launch_rows * 768 + 2 * B * 768 * 4 + 2 * B * k * 12
    s.close()
    return {"workload": f"configs[1] shape over the anisotropic (BERT-like) dump: {n} rows, batch {B}, top-{k}, L {L}",
            "queries_per_sec": B / dt, "ms_per_batch": dt * 1e3, "steps": steps,
            "uncertified_rows_all_timed_steps": n_uncert, "certified_by_first_attempt_four_batches": f"{fast}/{4 * 2 * B}",
            "scan_pairs_last_launch": pairs, "scan_emit_triggers_last_launch": triggers,
            "aux_layout": {"row_bytes": int(lay[0]), "norm_slots": int(lay[1]), "replica_slots": int(lay[2]),
                           "rogue_dims": sorted(set(int(d) for d in lay[4:4 + int(lay[2])]))},
            "finalize_seconds": fin_s, "recall_at_10_rows_checked": int(sel.numel()), "recall_at_10": rec10, "rows_with_identical_ids": rec,
            "encoder_like_queries": enc,
            "roofline": {"bound": "hbm", "kernel": f"dph_scan_kernel<1, 4, false, 0, 1, {'true' if lay[0] else 'false'}>", "achieved": alg / (avg / 1e3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (avg / 1e3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": avg, "launches": scan_n,
                         "algorithmic_bytes_per_launch": alg, "aux_bytes_per_launch": launch_rows * int(lay[0]), "traffic": None}}


def also_ivf(args, dev, local):
    """configs[3] on one GPU: the SURVEY 8d mixture dump (4096 Gaussians; kind 3), spherical k-means lists trained on a
    sample of the resident rows, every row assigned with the fused MFMA GEMM + arg-max, the list-major shard BUILT in HBM
    (radix sort + gather), then batch 256 through IVF-4096 / nprobe 256 (unit scan) and through the exact search of the same
    shard; recall of IVF against exact in-run; roofline on the bytes of the PROBED lists (SURVEY 8d "IVF scan")."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.ivf import assign_lists_resident, train_centroids_resident
    from densephrases_amd.synth import synthetic_rows
    n, nlist, nprobe, B, k, kind = 170_000_000 // 32 * 32, 4096, 256, 256, args.top_k, 3
    st = torch.cuda.current_stream(dev).cuda_stream
    s = Shard(n, device=local)
    s.fill_synthetic(seed=args.seed, kind=kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cent, _ = train_centroids_resident(s, nlist, iters=10, return_info=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    assign = assign_lists_resident(s, cent)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    s.make_list_major(assign.data_ptr(), cent, stream=st)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    counts = torch.bincount(assign.to(torch.int64), minlength=nlist)
    del assign
    torch.cuda.empty_cache()
    s.finalize()
    torch.cuda.synchronize()
    R = 2 * B
    rng = np.random.default_rng(7)
    rows = rng.integers(0, n, R)
    base = np.concatenate([synthetic_rows(int(r), 1, seed=args.seed, kind=kind) for r in rows]).astype(np.float32) / 20.0 - 2.0
    x = torch.from_numpy((base + rng.normal(0, 0.25, base.shape)).astype(np.float32)).to(dev)
    D = torch.empty((R, k), dtype=torch.float32, device=dev)
    I = torch.empty((R, k), dtype=torch.int64, device=dev)
    status = torch.empty(R, dtype=torch.int32, device=dev)

    def timed(fn, steps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s.profile_enable(True)
        s.profile_read()
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        scan_ms, _ = s.profile_read()
        return dt, scan_ms / steps, int((status == 0).sum().item()), s.stats()

    ivf = timed(lambda: s.search_ivf_dev(x.data_ptr(), R, k, nprobe, D.data_ptr(), I.data_ptr(), status.data_ptr()), 6)
    I_ivf = I.clone()
    exact = timed(lambda: s.search_dev(x.data_ptr(), R, k, D.data_ptr(), I.data_ptr(), status.data_ptr()), 2)
    rec = {}
    for kk in (1, 5, k):
        hit = (I_ivf[:, :kk, None] == I[:, None, :kk]).any(1).float().sum(1) / kk
        rec[f"recall_at_{kk}_vs_exact"] = float(hit.mean().item())
    # bytes of the probed lists: the probe set recomputed in torch (fp32 scores; a near-tie at the nprobe-th place may
    # swap one list), every list padded to whole 32-row tiles as it is stored
    probe = torch.topk(x @ torch.from_numpy(cent).to(dev).T, nprobe, dim=1).indices
    hit_lists = torch.zeros(nlist, dtype=torch.bool, device=dev)
    hit_lists[probe.flatten()] = True
    padded = (counts + 31) // 32 * 32
    probed_bytes = float((padded * hit_lists).sum().item()) * 768
    row_bytes = float((padded[probe.flatten()]).sum().item()) * 768         # what a per-query-row scan would read
    if not args.no_check:
        assert ivf[2] == R and exact[2] == R, (ivf[2:], exact[2:])
    out = {"workload": "configs[3] on 1 GPU: IVF-4096 (k-means lists built in HBM over the 170 M-row mixture dump), nprobe 256, "
                       "batch 256 (512 query rows), exact in-list inner product, top-10",
           "queries_per_sec": B / ivf[0], "ms_per_batch": ivf[0] * 1e3, "full_scan_ms_per_batch": ivf[1],
           "certified_rows": f"{ivf[2]}/{R}", "stats_last_call": ivf[3],
           "exact_same_shard": {"queries_per_sec": B / exact[0], "ms_per_batch": exact[0] * 1e3,
                                "full_scan_ms_per_batch": exact[1], "certified_rows": f"{exact[2]}/{R}"},
           **rec,
           "build_seconds": {"kmeans": t1 - t0, "assign": t2 - t1, "list_builder": t3 - t2},
           "lists": {"largest": int(counts.max().item()), "smallest": int(counts.min().item()),
                     "probed": int(hit_lists.sum().item()), "of": nlist},
           "roofline": {"bound": "hbm", "kernel": "dph_scan_units_kernel<0, false>", "achieved": probed_bytes / (ivf[1] / 1e3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": probed_bytes / (ivf[1] / 1e3) / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_batch": probed_bytes,
                        "bytes_if_every_query_row_scanned_its_own_lists": row_bytes,
                        "note": "bytes of the lists probed by >= 1 of the 512 rows, each read once per batch (SURVEY 8d)"}}
    s.profile_enable(False)
    s.close()
    del x, D, I, status, I_ivf, probe, hit_lists, padded, counts
    torch.cuda.empty_cache()
    try:
        out["recall_vs_nprobe_docruns"] = ivf_recall_sweep(args, dev, local)
    except Exception as e:                               # the sweep must not lose the configs[3] numbers above
        out["recall_vs_nprobe_docruns"] = {"error": repr(e)[:300]}
    return out


def ivf_recall_sweep(args, dev, local):
    """Where IVF recall is a real trade-off (the SURVEY 8d mixture above is so well separated that every nprobe finds everything): the
    DOCUMENT-ORDERED dump (runs of 56..200 near-duplicate rows, ~330 k runs over 4096 k-means lists built in HBM), a quarter of the
    configs[1] size to keep the leg short; batch 256, queries = a stored row + N(0, 0.25^2); recall@1/5/10 of IVF against the exact
    search of the same shard per nprobe.  (The kernel's answer equals the float64 IVF oracle id for id whatever the recall:
    tests/test_ivf.py::test_ivf_equals_the_oracle_on_2M_document_ordered_rows...; the reference's own IVF has the same recall.)"""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.ivf import make_list_major_resident
    from densephrases_amd.synth import synthetic_rows
    n, nlist, B, k, kind = 42_500_000 // 32 * 32, 4096, 256, args.top_k, 2
    s = Shard(n, device=local)
    s.fill_synthetic(seed=args.seed, kind=kind)
    t0 = time.perf_counter()
    make_list_major_resident(s, nlist, iters=10)
    s.finalize()
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    R = 2 * B
    rng = np.random.default_rng(17)
    rows = rng.integers(0, n, R)
    base = np.concatenate([synthetic_rows(int(r), 1, seed=args.seed, kind=kind) for r in rows]).astype(np.float32) / 20.0 - 2.0
    x = torch.from_numpy((base + rng.normal(0, 0.25, base.shape)).astype(np.float32)).to(dev)
    D = torch.empty((R, k), dtype=torch.float32, device=dev)
    I = torch.empty((R, k), dtype=torch.int64, device=dev)
    status = torch.empty(R, dtype=torch.int32, device=dev)
    s.search_dev(x.data_ptr(), R, k, D.data_ptr(), I.data_ptr(), status.data_ptr())
    torch.cuda.synchronize()
    I_exact = I.clone()
    ok_exact = int((status == 0).sum().item())
    sweep = []
    for nprobe in (1, 4, 16, 64, 256):
        for _ in range(2):
            s.search_ivf_dev(x.data_ptr(), R, k, nprobe, D.data_ptr(), I.data_ptr(), status.data_ptr())
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            s.search_ivf_dev(x.data_ptr(), R, k, nprobe, D.data_ptr(), I.data_ptr(), status.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        rec = {}
        for kk in (1, 5, k):
            hit = (I[:, :kk, None] == I_exact[:, None, :kk]).any(1).float().sum(1) / kk
            rec[f"recall_at_{kk}"] = float(hit.mean().item())
        sweep.append({"nprobe": nprobe, "ms_per_batch": dt * 1e3, "queries_per_sec": B / dt, "certified_rows": int((status == 0).sum().item()), **rec})
    s.close()
    at256 = sweep[-1]
    return {"dump": f"document-ordered runs (kind 2), {n} rows, IVF-{nlist} built in HBM ({build_s:.1f} s), batch {B}", "exact_certified_rows": ok_exact,
            "sweep": sweep,
            "north_star_criterion": {"recall_at_1_within_0.1_of_exact_at_nprobe_256": at256["recall_at_1"] >= 0.9,
                                     "recall_at_10_within_0.1_of_exact_at_nprobe_256": at256[f"recall_at_{k}"] >= 0.9}}


def measure_traffic(args, kernel):
    """HBM bytes per launch of the dominant kernel from the PMC counters, as MI355X_MICROARCH.md's HBM section prescribes:
    a SEPARATE pass of the same workload under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (2 timed steps; counters
    serialise the kernels, so nothing of this pass is timed), FETCH_SIZE is in KiB and on gfx950 counts a 128-B request
    as 64 B: bytes = avg(FETCH_SIZE) * 1024 * 2.  Returns (bytes or None, note)."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="dph_pmc_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--pmc", "FETCH_SIZE", "-d", tmp, "--", sys.executable, os.path.abspath(__file__), "--steps", "2",
           "--warmup", "1", "--batch", str(args.batch), "--top_k", str(args.top_k), "--dist", args.dist, "--seed", str(args.seed),
           "--no_cpu_baseline", "--no_also", "--no_traffic", "--recall_queries", "0"] + (["--rows", str(args.rows)] if args.rows else [])
    for t in args.tune:
        cmd += ["--tune", t]
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return None, f"rocprofv3 pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"
        cur = sqlite3.connect(dbs[0]).cursor()
        row = cur.execute("select avg(value), count(*) from counters_collection where counter_name = 'FETCH_SIZE' and "
                          "kernel_name like ?", ("%" + kernel + "%",)).fetchone()
        if not row or not row[1]:
            return None, "no FETCH_SIZE sample of " + kernel
        return float(row[0]) * 1024.0 * 2.0, (f"rocprofv3 --kernel-trace --pmc FETCH_SIZE over a separate 2-step pass of this workload: avg of "
                                             f"{row[1]} launches of {kernel}, KiB x 1024 x 2 (gfx950 counts a 128-B request as 64 B)")
    except (subprocess.TimeoutExpired, sqlite3.Error, OSError) as e:
        return None, "rocprofv3 pass: " + repr(e)[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pq_independent_topk(x, A, cent, pqc, block, sizes, nprobe, k, dev):
    """The answer of the PQ leg recomputed in plain torch, float64, no libdph code: x' = A x, the nprobe best lists by <x', c>, every code
    of those lists scored as <x', c_list> + sum_m <x'_m, codeword[m][code_m]> = <x', reconstruct(id)> (the synthetic index is a function of
    its seed: code of position p = block[(p % 2^20 - (p >> 20) % 97) mod 2^20], id = p).  Returns (scores [n,k], ids [n,k])."""
    import torch
    n = x.shape[0]
    M, ksub, dsub = pqc.shape
    xp = x.to(torch.float64) @ torch.from_numpy(A).to(dev).to(torch.float64).T
    nlist = cent.shape[0]
    coarse = torch.empty((n, nlist), dtype=torch.float64, device=dev)
    for c0 in range(0, nlist, 1 << 16):
        coarse[:, c0:c0 + (1 << 16)] = xp @ torch.from_numpy(cent[c0:c0 + (1 << 16)]).to(dev).to(torch.float64).T
    probe_s, probe = torch.topk(coarse, nprobe, dim=1)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)).to(dev)
    blk = torch.from_numpy(block).to(dev)
    pq = torch.from_numpy(pqc).to(dev).to(torch.float64)
    nb = blk.shape[0]
    out_s = torch.full((n, k), -float("inf"), dtype=torch.float64, device=dev)
    out_i = torch.full((n, k), -1, dtype=torch.int64, device=dev)
    for r in range(n):
        lens = off[probe[r] + 1] - off[probe[r]]
        tot = int(lens.sum().item())
        if tot == 0:
            continue
        first = torch.cumsum(lens, 0) - lens
        pos = torch.repeat_interleave(off[probe[r]], lens) + (torch.arange(tot, device=dev) - torch.repeat_interleave(first, lens))
        codes = blk[(pos % nb - (pos // nb) % 97) % nb].to(torch.int64)                        # [tot, M]
        table = torch.einsum("mt,mjt->mj", xp[r].reshape(M, dsub), pq)                        # [M, 256] float64
        sc = torch.repeat_interleave(probe_s[r], lens) + table.gather(1, codes.T).sum(0)
        kk = min(k, tot)
        order = torch.argsort(sc, descending=True, stable=True)[:kk]                          # positions ascend inside a list; ties are measure-zero here
        out_s[r, :kk], out_i[r, :kk] = sc[order], pos[order]
    return out_s, out_i


def pq_cpu_baseline(args, n, nlist, nprobe):
    """The reference's ACTUAL CPU path for its released index (index.py:53,62,200: IVFPQ, nprobe 256) restated with numpy on the host
    cores over the same synthetic index, in a process of its own (oracle/cpu_baseline_pq.py)."""
    r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline_pq", "--batch", str(args.batch), "--top_k", str(args.top_k), "--nlist", str(nlist),
                        "--codes", str(n), "--nprobe", str(nprobe), "--budget", "15"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        return {"error": "cpu IVFPQ baseline failed: " + r.stderr[-300:]}
    m = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": m["qps"], "unit": "queries/sec", "cores": m["cores"], "kind": "port", "seconds_per_batch": m["seconds_per_batch"],
            "sample": (f"oracle/cpu_baseline_pq.py: OPQ transform + IndexFlatIP coarse quantizer over all {nlist} centroids (blocked sgemm, one thread per "
                       f"block) + top-{nprobe} per row + ADC over the probed lists (one thread per query row, fp32 sequential sum like FAISS' scalar scan) "
                       f"with numpy on {m['cores']} threads over the SAME synthetic index, whole batches of {args.batch}: median of {m['batches']} "
                       f"batches; a port -- FAISS' SIMD scan kernels are not here")}


def also_pq(args, dev, local):
    """The reference's own index type at the released index's shape (model.py:18 `1048576_flat_OPQ96`): OPQ96 + IVFPQ with 2^20
    lists, nprobe 256 (index.py:53), 170 M codes resident in HBM (synthetic codes: parity is tests/test_pq.py), batch 64 through
    dph_search_ivf_dev: OPQ transform, coarse quantizer (one-product bf16 filter GEMM over 2^20 centroids with the threshold test in
    its epilogue, float64 re-rank of the error band), LUT-in-LDS ADC scan grouped by query row, exact top-k.  In-run: the answer of
    16 query rows recomputed independently in plain torch (float64), roofline of the dominant kernel (the filter GEMM: its HIP-event
    time against the bytes of the bf16 centroid matrix), and the CPU baseline of the SAME search on the host cores."""
    import torch
    from densephrases_amd.synth import synthetic_pq_shard
    n, nlist, nprobe, B, k = 170_000_000, 1 << 20, 256, args.batch, args.top_k
    t0 = time.perf_counter()
    s, A, cent, sizes, pqc, block = synthetic_pq_shard(n, nlist, 96, device=local, return_parts=True, doc_len=100)
    torch.cuda.synchronize()
    load_s = time.perf_counter() - t0
    R = 2 * B
    x = torch.from_numpy(np.random.default_rng(3).normal(0, 0.5, (R, 768)).astype(np.float32)).to(dev)
    D = torch.empty((R, k), dtype=torch.float32, device=dev)
    I = torch.empty((R, k), dtype=torch.int64, device=dev)
    st = torch.empty(R, dtype=torch.int32, device=dev)
    fn = lambda: s.search_ivf_dev(x.data_ptr(), R, k, nprobe, D.data_ptr(), I.data_ptr(), st.data_ptr())     # noqa: E731
    s.profile_enable(True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s.profile_read()
    steps = 20
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    gemm_ms, gemm_n = s.profile_read()
    failed_over, emitted = s.debug_pq_coarse()
    ok = int((st == 0).sum().item())
    assert args.no_check or (ok == R and int((I[:, 0] >= 0).sum().item()) == R), ok
    # ---- independent check: 16 query rows again in plain torch, float64
    n_chk = min(16, R)
    sel = torch.linspace(0, R - 1, n_chk).round().to(torch.int64).to(dev)
    ref_s, ref_i = pq_independent_topk(x[sel], A, cent, pqc, block, sizes, nprobe, k, dev)
    got_i, got_d = I[sel], D[sel].to(torch.float64)
    ids_equal = int((got_i == ref_i).all(1).sum().item())
    # an id the reference does not list may only be there on a near-tie (the kernel sums in fp32): its score must be within fp32 noise of the k-th
    stray_ok = True
    for r in range(n_chk):
        extra = [j for j in range(k) if int(got_i[r, j]) not in set(ref_i[r].tolist())]
        for j in extra:
            stray_ok = stray_ok and abs(float(got_d[r, j]) - float(ref_s[r, k - 1])) <= 2e-5 * max(1.0, abs(float(ref_s[r, k - 1])))
    rel = float(((got_d - ref_s).abs() / ref_s.abs().clamp_min(1.0)).max().item())
    assert args.no_check or (stray_ok and rel < 2e-5 and ids_equal >= n_chk - 2), (ids_equal, rel, stray_ok)
    probe = torch.topk((x @ torch.from_numpy(A).to(dev).T) @ torch.from_numpy(cent).to(dev).T, nprobe, dim=1).indices
    scanned = float(torch.from_numpy(sizes).to(dev)[probe.flatten()].sum().item())
    try:
        e2e = pq_e2e(s, args, dev, B / dt)
    except Exception as e:                                  # the search-only numbers above must survive a failing sub-leg
        e2e = {"error": repr(e)[:300]}
    try:
        b512 = pq_b512_document(s, args, dev)
    except Exception as e:
        b512 = {"error": repr(e)[:300]}
    s.close()
    del ref_s, ref_i, probe
    torch.cuda.empty_cache()
    out = {"workload": f"IndexPreTransform(OPQ96) -> IndexIVFPQ, 2^20 lists, nprobe {nprobe}, {n} codes in HBM, batch {B} ({R} query rows), top-{k}",
           "queries_per_sec": B / dt, "ms_per_batch": dt * 1e3, "steps": steps, "exact_rows": f"{ok}/{R}", "codes_scored_per_batch": scanned,
           "coarse_failed_over_last_batch": failed_over, "coarse_candidates_per_row": emitted / R,
           "independent_check": {"rows": n_chk, "rows_with_identical_ids": ids_equal, "max_rel_score_diff": rel,
                                 "how": "plain torch, float64: x' = A x, top-256 lists by <x', c>, <x', reconstruct(id)> over every code of those lists"},
           "index_load_seconds": load_s, "e2e_mips_search": e2e, "b512_document_stream": b512}
    if gemm_n:
        gemm_s = gemm_ms / gemm_n / 1e3
        alg = nlist * 768 * 2 + R * 768 * 2 + emitted * 10               # the bf16 centroid matrix once + the query image + the candidates out
        flop = 2.0 * R * 768 * nlist
        out["roofline"] = {"bound": "hbm", "kernel": "dph_coarse_scan_kernel (each launch alone inside its event pair; a rocprofv3 dispatch runs ~10 us shorter: profiles/r06_trace_pq.json)", "achieved": alg / gemm_s / 1e9, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": alg / gemm_s / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": gemm_s * 1e3,
                           "launches": gemm_n, "algorithmic_bytes_per_launch": alg,
                           "share_of_batch": gemm_s / dt,
                           "mfma_bf16": {"achieved": flop / gemm_s / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": flop / gemm_s / 1e12 / 2500.0}}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = pq_cpu_baseline(args, n, nlist, nprobe)
    return out


def pq_b512_document(s, args, dev):
    """configs[4] over the index type the reference serves it from (Makefile:347-375 KILT runs -> model.py:79-87): batch 512 streaming
    queries, retrieval_unit='document' => search_top_k = 2 * top_k, agg_strat opt3 (title de-duplication), MIPS.search_stream over the
    OPQ96 / 2^20-list index (index.py:391-448 the host half).  Reports the search alone (dph_search_ivf_dev, 1024 query rows), the GPU
    half of a step (search + both pq_window passes), the host half, what of it the stream does not hide, and queries/sec."""
    import gc
    import torch
    from densephrases_amd import MIPS
    from densephrases_amd.dist import ShardedSearcher
    from densephrases_amd.synth import SynthDocStore
    B, k, L, nprobe = 512, 2 * args.top_k, args.max_answer_length, 256
    R = 2 * B
    x = torch.from_numpy(np.random.default_rng(23).normal(0, 0.5, (R, 768)).astype(np.float32)).to(dev)
    D = torch.empty((R, k), dtype=torch.float32, device=dev)
    I = torch.empty((R, k), dtype=torch.int64, device=dev)
    st = torch.empty(R, dtype=torch.int32, device=dev)
    fn = lambda: s.search_ivf_dev(x.data_ptr(), R, k, nprobe, D.data_ptr(), I.data_ptr(), st.data_ptr())     # noqa: E731
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s.profile_read()
    steps = 10
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    search_ms = (time.perf_counter() - t) / steps * 1e3
    coarse_ms, coarse_n = s.profile_read()
    ok = int((st == 0).sum().item())
    mips = MIPS.from_shard(s, SynthDocStore())
    rng = np.random.default_rng(29)
    batches = [rng.normal(0, 0.5, (B, 1536)).astype(np.float32) for _ in range(4)]
    kw = dict(top_k=k, aggregate=True, agg_strat="opt3", max_answer_length=L)
    gc.collect()
    gc.freeze()
    for _ in mips.search_stream((batches[i % 4] for i in range(5)), **kw):       # (also fetches the four batches' documents into the host half's cache)
        pass
    tm = mips.reset_timing()
    steps = 12
    t0 = time.perf_counter()
    n_res = n_out = 0
    for outs in mips.search_stream((batches[i % 4] for i in range(steps)), q_texts=(["q"] * B for _ in range(steps)), **kw):
        n_out += len(outs)
        n_res += sum(len(r) for r in outs)
    dt = (time.perf_counter() - t0) / steps
    host_ms, wait_ms, enq_ms = (tm[key] / steps * 1e3 for key in ("host_s", "wait_s", "enqueue_s"))
    ss = ShardedSearcher(s, B, k, L, device=dev)
    qd = [torch.from_numpy(b).to(dev) for b in batches]
    for i in range(2):
        ss.step(qd[i % 4])
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(steps):
        ss.step(qd[i % 4])
    ev1.record()
    torch.cuda.synchronize()
    dev_ms = ev0.elapsed_time(ev1) / steps
    gc.unfreeze()
    assert args.no_check or (ok == R and n_out == steps * B and n_res > 0), (ok, n_out, n_res)
    return {"workload": f"configs[4] over the OPQ96-IVFPQ index: batch {B} ({R} query rows), retrieval_unit=document (top_k doubled to {k}, agg_strat opt3), "
                        "MIPS.search_stream: host queries in, de-duplicated result dicts out",
            "queries_per_sec": B / dt, "ms_per_batch": dt * 1e3, "steps": steps,
            "search_only_ms_per_batch": search_ms, "search_only_queries_per_sec": B / (search_ms / 1e3),
            "coarse_filter_ms_per_batch": coarse_ms / 10.0 if coarse_n else None,       # (the event pair of every pass: sample GEMM excluded, scan + bucket launches of all row groups)
            "device_ms_per_batch": dev_ms, "host_ms_per_batch": host_ms, "enqueue_ms_per_batch": enq_ms, "gpu_wait_ms_per_batch": wait_ms,
            "host_id2docword_ms_per_batch": tm.get("host_idx_s", 0.0) / steps * 1e3, "host_assemble_ms_per_batch": tm.get("host_assemble_s", 0.0) / steps * 1e3,
            "exposed_host_ms": max(0.0, dt * 1e3 - dev_ms), "stream_over_search_only": (B / dt) / (B / (search_ms / 1e3)),
            "results_per_query": n_res / max(n_out, 1), "host_threads": int(os.environ.get("DPH_HOST_THREADS", "0")) or min(8, (os.cpu_count() or 2) // 2),
            "host_us_per_candidate": host_ms * 1e3 / (2 * B * k)}


def also_pq_skewed(args, dev, local):
    """the PQ search over an index whose list sizes are SKEWED like a k-means quantizer's over token vectors
    (build_phrase_index.py:113-116,156-279): zipf sizes (the longest list ~800 k codes = 5000 x the mean, a hundred above 30 k) whose
    centroids are also the longest, i.e. the most probed under inner product (synth.synthetic_pq_parts).  The scan's work is cut by
    code count (units of <= 12288 codes), so what a batch costs follows the codes it has to score: reported with the codes per batch
    and the time per code next to the near-uniform index's, the answer of 8 rows recomputed independently in float64.  `giant`: the
    near-uniform index with ONE 600 k-code list that 2 % of the rows probe (the r05 pathology: 3.1 ms per batch of 256 behind one
    workgroup)."""
    import torch
    from densephrases_amd.synth import synthetic_pq_shard
    n, nlist, nprobe, B, k = 170_000_000, 1 << 20, 256, args.batch, args.top_k
    out = {"workload": f"IndexPreTransform(OPQ96) -> IndexIVFPQ, 2^20 lists of zipf sizes (long lists = long centroids = most probed), nprobe {nprobe}, {n} codes, batch {B}"}
    for skew in ("zipf", "giant"):
        s, A, cent, sizes, pqc, block = synthetic_pq_shard(n, nlist, 96, device=local, return_parts=True, skew=skew)
        rec = {}
        for Bq in (B, 256):
            R = 2 * Bq
            x = torch.from_numpy(np.random.default_rng(3).normal(0, 0.5, (R, 768)).astype(np.float32)).to(dev)
            D = torch.empty((R, k), dtype=torch.float32, device=dev)
            I = torch.empty((R, k), dtype=torch.int64, device=dev)
            st = torch.empty(R, dtype=torch.int32, device=dev)
            fn = lambda: s.search_ivf_dev(x.data_ptr(), R, k, nprobe, D.data_ptr(), I.data_ptr(), st.data_ptr())     # noqa: E731
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            steps = 10
            t = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / steps
            ok = int((st == 0).sum().item())
            failed_over, _ = s.debug_pq_coarse()
            probe = torch.topk((x @ torch.from_numpy(A).to(dev).T) @ torch.from_numpy(cent).to(dev).T, nprobe, dim=1).indices
            psz = torch.from_numpy(sizes).to(dev)[probe]
            scanned = float(psz.sum().item())
            r = {"queries_per_sec": Bq / dt, "ms_per_batch": dt * 1e3, "exact_rows": f"{ok}/{R}", "coarse_failed_over": failed_over,
                 "codes_scored_per_batch": scanned, "ns_per_code_per_workgroup": dt * 1e9 * 256 / scanned,
                 "rows_probing_the_longest_list": int((probe == int(np.argmax(sizes))).any(1).sum().item())}
            assert args.no_check or ok == R, (skew, Bq, ok)
            if Bq == B and skew == "zipf":
                sel = torch.linspace(0, R - 1, 8).round().to(torch.int64).to(dev)
                ref_s, ref_i = pq_independent_topk(x[sel], A, cent, pqc, block, sizes, nprobe, k, dev)
                got_i, got_d = I[sel], D[sel].to(torch.float64)
                rel = float(((got_d - ref_s).abs() / ref_s.abs().clamp_min(1.0)).max().item())
                same = int((got_i == ref_i).all(1).sum().item())
                assert args.no_check or (rel < 2e-5 and same >= 6), (rel, same)
                r["independent_check"] = {"rows": 8, "rows_with_identical_ids": same, "max_rel_score_diff": rel}
            rec[f"b{Bq}"] = r
        srt = np.sort(sizes)[::-1]
        rec["list_sizes"] = {"mean": float(sizes.mean()), "top5": [int(v) for v in srt[:5]], "over_100x_mean": int((sizes > 100 * sizes.mean()).sum())}
        s.close()
        del s, A, cent, sizes, pqc, block
        torch.cuda.empty_cache()
        if skew == "zipf":
            out.update({"queries_per_sec": rec[f"b{B}"]["queries_per_sec"], "ms_per_batch": rec[f"b{B}"]["ms_per_batch"], **rec})
        else:
            out["giant"] = rec
    return out


def pq_e2e(s, args, dev, search_only_qps):
    """VERDICT r4 "missing" 1: the reference's SHIPPING configuration end to end -- MIPS.search / search_stream over the OPQ96 / 2^20-list
    index with idx2id + f2o attached (index.py:276-302 the PQ branch's reconst_fn per candidate x L, :323-370 the window re-score,
    :391-448 dict assembly + aggregate_results; model.py:18,82-87), B = 64, k = 10, L = 10, aggregate=True: host float queries in,
    result dicts out.  `device_ms_per_batch`: the GPU half alone (search + both pq_window passes, HIP-timed through
    ShardedSearcher.step), host_ms: the C++ host half + python dict creation, exposed = what the stream does not hide."""
    import torch
    from densephrases_amd import MIPS
    from densephrases_amd.dist import ShardedSearcher
    from densephrases_amd.synth import SynthDocStore
    B, k, L = args.batch, args.top_k, args.max_answer_length
    mips = MIPS.from_shard(s, SynthDocStore())
    rng = np.random.default_rng(17)
    batches = [rng.normal(0, 0.5, (B, 1536)).astype(np.float32) for _ in range(4)]
    texts = ["q"] * B
    kw = dict(top_k=k, aggregate=True, agg_strat="opt1", max_answer_length=L)
    steps, warm = 20, 6
    # This leg runs late in a long-lived process: the earlier legs left millions of live container objects behind (document caches,
    # result lists), and ONE full cyclic collection over them costs ~0.2 s -- 11 ms per batch when it lands in a 20-batch timed region
    # (seen in the first r05 run: host_ms_per_batch 11.9 here against 0.8 in tools/pq_e2e.py, a fresh process).  What a serving
    # process does after its warm-up: collect once, freeze the survivors out of the collector's reach.
    import gc
    gc.collect()
    gc.freeze()
    full_gc0 = gc.get_stats()[2]["collections"]
    for i in range(warm):                                   # (also fetches the documents of the four batches into the host half's cache)
        out = mips.search(batches[i % 4], q_texts=texts, **kw)
    t0 = time.perf_counter()
    for i in range(steps):
        out = mips.search(batches[i % 4], q_texts=texts, **kw)
    dt = (time.perf_counter() - t0) / steps
    n_res = sum(len(r) for r in out)
    for _ in mips.search_stream((batches[i % 4] for i in range(4)), **kw):
        pass
    tm = mips.reset_timing()
    t0 = time.perf_counter()
    n_out = 0
    for outs in mips.search_stream((batches[i % 4] for i in range(steps)), q_texts=(texts for _ in range(steps)), **kw):
        n_out += len(outs)
    dt_s = (time.perf_counter() - t0) / steps
    host_ms, wait_ms, enq_ms = (tm[key] / steps * 1e3 for key in ("host_s", "wait_s", "enqueue_s"))
    # the GPU half alone
    ss = ShardedSearcher(s, B, k, L, device=dev)
    qd = [torch.from_numpy(b).to(dev) for b in batches]
    for i in range(3):
        ss.step(qd[i % 4])
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(steps):
        ss.step(qd[i % 4])
    ev1.record()
    torch.cuda.synchronize()
    dev_ms = ev0.elapsed_time(ev1) / steps
    full_gcs = gc.get_stats()[2]["collections"] - full_gc0
    gc.unfreeze()
    assert args.no_check or (n_out == steps * B and n_res > 0), (n_out, n_res)
    return {"workload": f"MIPS.search / search_stream over the OPQ96-IVFPQ index with idx2id + f2o: host queries in, aggregated result dicts out, batch {B}, top-{k}, L {L}",
            "queries_per_sec": B / dt, "ms_per_batch": dt * 1e3,
            "search_stream_queries_per_sec": B / dt_s, "search_stream_ms_per_batch": dt_s * 1e3,
            "device_ms_per_batch": dev_ms, "device_only_queries_per_sec": B / (dev_ms / 1e3),
            "host_ms_per_batch": host_ms, "enqueue_ms_per_batch": enq_ms, "gpu_wait_ms_per_batch": wait_ms,
            "exposed_host_ms": max(0.0, dt_s * 1e3 - dev_ms),
            "stream_over_search_only": (B / dt_s) / search_only_qps, "results_last_batch": n_res, "steps": steps,
            "full_gc_collections_during_leg": int(full_gcs)}


def make_line(args, world, weak, n_total, n_local, elapsed, scan_ms, scan_launches, ladder_ms, ladder_launches, stats, pairs,
              triggers, n_uncert):
    """The JSON line of the headline measurement from what the timed region measured (pure arithmetic: tests/test_bench_contract.py
    calls it with made-up measurements).  Returns (line, name of the dominant kernel, algorithmic bytes per launch)."""
    B, k, L = args.batch, args.top_k, args.max_answer_length
    n_rows_q = 2 * B
    passes = []                           # (rows of the pass, qb) exactly as dph_search_dev cuts the batch
    left = n_rows_q
    while left > 0:
        qb = 2 if left > 128 else 1
        passes.append(min(left, 128 * qb))
        left -= passes[-1]
    launches_per_step = len(passes)
    avg_scan_s = scan_ms / max(scan_launches, 1) / 1e3
    rows_per_launch = n_rows_q / launches_per_step
    # bytes the full-scan LAUNCH has to read: with the finest ladder level (every S-th tile) fused into it, the launch visits the
    # other tiles only -- the level's tiles are read by the level's own launch, i.e. the dump once per batch
    fused = int(stats.get("fused_stride", 0) or 0)
    tiles = (n_local + 31) // 32
    launch_rows = (tiles - (tiles + fused - 1) // fused) * 32 if fused >= 2 else n_local
    alg_launch = launch_rows * 768 * 1 + rows_per_launch * 768 * 4 + rows_per_launch * k * 12   # SURVEY.md 8(d), s = 1
    achieved = alg_launch / avg_scan_s / 1e9
    # per BATCH: the whole local dump once + the queries + the results, over ALL the HBM-bound scan launches of the
    # batch -- the full scans AND the ladder levels' sampled scans (a batch of more than 256 query rows reads the dump once per pass:
    # that shows up as a lower fraction here, the bytes stay the algorithm's)
    alg_batch = n_local * 768 * 1 + n_rows_q * 768 * 4 + n_rows_q * k * 12
    all_scan_s_per_step = (scan_ms + ladder_ms) / 1e3 / args.steps
    scan_s_per_step = scan_ms / 1e3 / args.steps
    qb_max = 2 if n_rows_q > 128 else 1
    mfma_ops = 2.0 * sum(128 * (2 if p > 128 else 1) for p in passes) * 768 * launch_rows       # int8 MACs*2 the timed full scans issue per step
    # the full-scan instantiation: <query groups per wave, staging sets, IVF masks, role 0 = full scan, hand-over schedule (tuning key scan_sched, default 1)>
    sched = 1
    for t in getattr(args, "tune", []) or []:
        if t.startswith("scan_sched="):
            vals = [int(v) for v in t.split("=", 1)[1].split(",") if v != ""]
            sched = vals[0] if qb_max == 1 else vals[-1]
    # ... and whether the shard carries aux rows (libdph found rogue dimensions / heavy-tailed row norms: the aux k-step)
    aux_stride = int(stats.get("aux_stride", 0) or 0)
    kernel = f"dph_scan_kernel<{qb_max}, 4, false, 0, {sched}, {'true' if aux_stride else 'false'}>"
    if world > 1:
        config_name = ("configs[2] sizing (162.5 M rows per GPU, 1.3 B over 8)" if weak else
                       f"{n_total} rows range-partitioned over {world} GPUs (strong scaling)")
    else:
        config_name = {64: "configs[1]", 256: "configs[3] batch shape, exact search",
                       512: "configs[4] batch shape, exact search"}.get(B, f"configs[1] dump at batch {B}")
        if n_total != 170_000_000:
            config_name += f" ({n_total} rows)"
    ms_per_step = elapsed / args.steps * 1e3
    line = {
        "metric": "queries/sec", "value": args.steps * B / elapsed, "unit": "queries/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None, "dtype": "int8",
        "row_queries_per_sec": args.steps * B / elapsed * n_total,
        "uncertified_rows_all_timed_steps": n_uncert,
        "certified_by_first_attempt_last_step": f"{stats['certified_fast']}/{stats['rows']}",
        "scan_pairs_last_launch": pairs, "scan_emit_triggers_last_launch": triggers,
        "data": "synthetic",
        "config": {"workload": (f"{config_name}: brute-force exact IP top-k + start/end window re-score, batch {B} "
                                f"({n_rows_q} query rows), int8 phrase dump resident in HBM"),
                   "rows_total": n_total, "rows_per_gpu": n_local, "dim": 768, "batch": B, "top_k": k,
                   "max_answer_length": L, "storage": "int8 (x = n/20 - 2)", "dump": args.dist,
                   "parallelism": f"range-shard x{world}", "scan_launches_per_step": launches_per_step},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "traffic_note": "not measured (--no_traffic / N > 1); rocprofv3 FETCH_SIZE passes are under profiles/",
                     "kernel": kernel, "avg_launch_ms": avg_scan_s * 1e3, "launches": scan_launches,
                     "fused_ladder_stride": fused, "rows_read_by_the_launch": launch_rows,
                     "algorithmic_bytes_per_launch": alg_launch,
                     "aux_row_bytes": aux_stride, "aux_bytes_per_launch": launch_rows * aux_stride,     # overhead, not algorithmic bytes
                     "per_batch": {"algorithmic_bytes": alg_batch, "scan_ms": all_scan_s_per_step * 1e3,
                                   "full_scan_ms": scan_s_per_step * 1e3, "ladder_scan_ms": ladder_ms / args.steps,
                                   "ladder_scan_launches": ladder_launches / args.steps,
                                   "achieved": alg_batch / all_scan_s_per_step / 1e9,
                                   "frac": alg_batch / all_scan_s_per_step / 1e9 / HBM_PEAK_GBS,
                                   "step_frac": alg_batch / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS,
                                   "note": "the whole local dump counted ONCE per batch (SURVEY 8d) over the summed time of ALL scan launches "
                                           "of the batch (full scans + ladder levels); step_frac: the same bytes over the whole step"},
                     "mfma_int8": {"achieved": mfma_ops / scan_s_per_step / 1e12, "peak": I8_MFMA_PEAK_TOPS,
                                   "unit": "TOP/s", "frac": mfma_ops / scan_s_per_step / 1e12 / I8_MFMA_PEAK_TOPS}},
        "fixed_ms_per_step": ms_per_step - scan_s_per_step * 1e3,
    }
    return line, kernel, alg_launch


def independent_topk(shard_rows_ptr, n_local, id_base, xq, k, dev):
    """fp64 brute force over the resident shard in plain torch (no libdph code): the reference answer for recall@k."""
    import torch
    db = torch.as_tensor(_DevRows(shard_rows_ptr, n_local), device=dev)
    q = xq.to(torch.float64)
    best_s = torch.full((q.shape[0], k), -float("inf"), dtype=torch.float64, device=dev)
    best_i = torch.full((q.shape[0], k), -1, dtype=torch.int64, device=dev)
    step = 1 << 20
    for r0 in range(0, n_local, step):
        # the reference's fp32 de-quantisation up to one ulp (torch divides by a scalar as t * fl(1/20) on the GPU): the ids
        # compared below are separated by far more than that
        xb = db[r0:r0 + step].to(torch.float32) / 20.0 - 2.0
        s = q @ xb.to(torch.float64).T
        ts, ti = torch.topk(s, min(k, s.shape[1]), dim=1)
        cs = torch.cat([best_s, ts], 1)
        ci = torch.cat([best_i, ti + (r0 + id_base)], 1)
        o = torch.topk(cs, k, dim=1)
        best_s, best_i = o.values, torch.gather(ci, 1, o.indices)
    return best_s, best_i


class _DevRows:
    """__cuda_array_interface__ view of the shard's resident rows (a raw device pointer owned by libdph)."""

    def __init__(self, ptr, n_rows):
        self.__cuda_array_interface__ = {"shape": (n_rows, 768), "typestr": "|i1", "data": (int(ptr), False), "version": 2}


def effective_cpus():
    """CPUs this process may really use: scheduler affinity capped by the cgroup CPU quota (v2 cpu.max / v1 cfs_quota_us)"""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, split in (("/sys/fs/cgroup/cpu.max", True), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", False)):
        try:
            with open(path) as f:
                txt = f.read().split()
            if split:
                if txt[0] != "max":
                    return max(1, min(n, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = int(f.read())
                if int(txt[0]) > 0 and per > 0:
                    return max(1, min(n, math.ceil(int(txt[0]) / per)))
        except (OSError, ValueError, IndexError):
            continue
    return n


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: launch the N ranks (one per GPU) and pass their output through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    t_start = time.perf_counter()
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    # the pod may use fewer CPUs than it sees (cgroup quota: 16 of 256 on the round-6 boxes): torch's intra-op pool defaults to one thread
    # per visible CPU, and that many spinning OpenMP threads eat the quota the host half of the end-to-end legs lives on
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    if args.trace_out:
        leg = "anisotropic" if args.dist == "anisotropic" else "flat"
        sys.exit(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_out.py"), "--leg", leg, "--out", args.trace_out, "--steps", str(args.steps),
                                 "--warmup", str(args.warmup)] + (["--rows", str(args.rows)] if args.rows else [])).returncode)
    import torch
    torch.set_num_threads(max(1, min(effective_cpus(), 16)))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    import __graft_entry__ as g
    g.build()
    one_gpu = world > 1 and os.environ.get("DPH_BENCH_ONE_GPU") == "1"       # rehearsal: every rank on cuda:0, gloo
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group(backend="gloo")
            from densephrases_amd.dist import HostStagedCollectives
            dist = HostStagedCollectives()
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    preflight = None
    if world > 1:
        # ---- preflight, BEFORE 125 GB per rank are filled: the process group is the one the driver asked for (RCCL, one distinct
        #      GPU per rank), and a small all-gather + all-reduce crosses it; rank 0 prints what it saw to stderr and keeps it for the line
        props = torch.cuda.get_device_properties(local)
        me = {"rank": rank, "local_rank": local, "device": int(torch.cuda.current_device()), "name": props.name,
              "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": int(getattr(props, "pci_bus_id", -1)),
              "hbm_gib": props.total_memory / (1 << 30), "pid": os.getpid()}
        seen = [None] * world
        if one_gpu:
            import torch.distributed as tdist
            tdist.all_gather_object(seen, me)
            backend = tdist.get_backend()
        else:
            dist.all_gather_object(seen, me)
            backend = dist.get_backend()
            assert backend == "nccl", f"bench.py --gpus {world}: backend {backend!r}, expected nccl (= RCCL on ROCm)"
            devs = {(d["pci_bus_id"], d["uuid"], d["device"]) for d in seen}
            assert len(devs) == world, f"{world} ranks on {len(devs)} distinct GPUs: {seen}"
            need = 162_500_000 * 768 / (1 << 30) * 1.08 if args.rows == 0 else 0
            assert all(d["hbm_gib"] > need for d in seen), f"a rank's GPU has less HBM than its {need:.0f} GiB shard: {seen}"
            t = torch.full((4,), float(rank + 1), device=dev)
            dist.all_reduce(t)                                 # the first RCCL collective of the run: fails here, not after the fill
            torch.cuda.synchronize()
            assert float(t[0].item()) == world * (world + 1) / 2, "all_reduce over RCCL returned a wrong sum"
        preflight = {"backend": backend, "world_size": world, "ranks_seen": len(seen),
                     "devices": [f"rank {d['rank']}: cuda:{d['device']} {d['name']} pci {d['pci_bus_id']}" for d in seen]}
        if rank == 0:
            print("[bench preflight] " + json.dumps(preflight), file=sys.stderr, flush=True)

    from densephrases_amd import Shard
    from densephrases_amd.dist import ShardedSearcher, partition_rows
    from densephrases_amd.synth import synthetic_rows

    B, k, L = args.batch, args.top_k, args.max_answer_length
    kind = {"iid": 0, "mixture": 1, "docruns": 2, "anisotropic": 4}[args.dist]
    weak = world > 1 and args.rows == 0
    n_total = args.rows or (170_000_000 if world == 1 else 162_500_000 * world)
    lo, hi = partition_rows(n_total, world)[rank]
    n_local = hi - lo
    shard = Shard(n_local, device=local, id_base=lo)
    shard.fill_synthetic(seed=args.seed, kind=kind)
    # synthetic idx2id / f2o: documents of 100 rows, every token kept (f2o = identity)
    doc = ((np.arange(n_local, dtype=np.int64) + lo) // 100).astype(np.int32)
    word = ((np.arange(n_local, dtype=np.int64) + lo) % 100).astype(np.int32)
    shard.set_idx2id(doc, word)
    d0, d1 = lo // 100, (hi + 99) // 100
    doc_ids = np.arange(d0, d1 + 1, dtype=np.int32)
    shard.set_f2o(doc_ids, np.arange(0, (len(doc_ids) + 1) * 100, 100, dtype=np.int64),
                  np.tile(np.arange(100, dtype=np.int32), len(doc_ids)))
    del doc, word
    shard.finalize()
    for t in args.tune:
        key, _, vals = t.partition("=")
        shard.set_tuning(key, *[int(v) for v in vals.split(",") if v != ""])

    searcher = ShardedSearcher(shard, B, k, L, rank=rank, world=world, dist=dist, device=dev)
    # queries: synthetic NQ-shaped batches, half of them planted near stored rows so the result is checkable
    # BASELINE.md section 3 / run_demo.py:329-352: every timed batch is a DIFFERENT batch of questions (1000 questions = 15 full batches of
    # 64), the warm-up batches are others again; at most 64 distinct batches are kept resident (longer runs cycle them)
    n_distinct = min(max(args.warmup, 1) + args.steps, 64)
    batches, planted = make_batches(args, B, n_total, kind, n_distinct, dev, style=args.queries)

    # the warm-up runs EXACTLY what a timed step runs (profiling events, the status reduction): the first use of any
    # kernel loads its code object, which must not land in the timed region
    shard.profile_enable(True)
    searcher.time_collectives = world > 1
    n_fail = torch.zeros((), dtype=torch.int64, device=dev)
    for i in range(max(args.warmup, 1)):
        out = searcher.step(batches[i % len(batches)])
        n_fail += (out["status"] != 0).sum()
    torch.cuda.synchronize()
    shard.profile_read()                    # discard the warm-up launches
    searcher.collective_ms()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    n_fail.zero_()                          # uncertified rows over ALL timed steps (device-side sum)
    torch.cuda.synchronize()
    # per-step durations for the median / minimum next to the mean: one event after every step on the
    # stream the steps run on -- nothing waits for them inside the timed region
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    step_ev[0].record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        out = searcher.step(batches[(max(args.warmup, 1) + i) % len(batches)])
        n_fail += (out["status"] != 0).sum()
        step_ev[i + 1].record()
        if args.per_step:
            torch.cuda.synchronize()
            print(f"step {i}: {(time.perf_counter() - ts) * 1e3:.2f} ms  {shard.stats()}", file=sys.stderr)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
    scan_ms, scan_launches, ladder_ms, ladder_launches = shard.profile_read_all()
    stats = shard.stats()                    # of the last step: how many rows the first attempt certified
    stats["aux_stride"] = int(shard.aux_layout()[0])
    pairs, triggers = shard.scan_counters()
    per_rank_ms = None
    if dist is not None:
        # per rank: its own step time, its full-scan launch (-> its own roofline fraction) and what its stream waited for in each of the
        # two exchanges (collective + the slowest rank's arrival) -- so that the first real SCALE run says WHERE the time went
        cm = searcher.collective_ms()
        tiles_l = (n_local + 31) // 32
        fused_l = int(stats.get("fused_stride", 0) or 0)
        rows_l = (tiles_l - (tiles_l + fused_l - 1) // fused_l) * 32 if fused_l >= 2 else n_local
        alg_l = rows_l * 768 + 2 * B * 768 * 4 + 2 * B * k * 12
        avg_l = scan_ms / max(scan_launches, 1)
        mine = torch.tensor([elapsed / args.steps * 1e3, avg_l, alg_l / (avg_l / 1e3) / 1e9 / HBM_PEAK_GBS if avg_l > 0 else 0.0,
                             cm["sample_all_gather"][0] / args.steps, cm["record_all_gather_and_merge"][0] / args.steps, float(n_local),
                             float(np.median(step_ms))], dtype=torch.float64, device=dev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        col = lambda j: [float(a[j].item()) for a in allr]     # noqa: E731
        per_rank_ms = {"ms_per_step": col(0), "avg_full_scan_ms": col(1), "roofline_frac": col(2), "sample_all_gather_wait_ms_per_step": col(3),
                       "record_all_gather_and_merge_wait_ms_per_step": col(4), "rows": [int(v) for v in col(5)], "ms_per_step_median": col(6),
                       "note": "the waits are event pairs on the step's stream around the exchange: the collective plus the arrival of the slowest rank"}
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity of the timed result: every row of every timed step certified, planted rows first (a wrong-but-fast run
    # must not produce a number)
    last = (max(args.warmup, 1) + args.steps - 1) % len(batches)
    I_all = out["I"].cpu().numpy()
    I_start = I_all[:B]
    n_uncert = int(n_fail.item())
    if not args.no_check:
        assert n_uncert == 0, f"uncertified rows in the timed region: {n_uncert}"
        if args.queries == "encoder":
            pass                              # (the mean of a run: any row of the run may come first; the brute-force comparison below applies)
        elif kind == 0:
            assert (I_start[:B // 2, 0] == planted[last]).all(), "planted rows did not come back first"
        elif kind != 4:
            # mixture dump: the saturated outlier rows legitimately out-score a planted row for some queries (inner
            # product search favours large norms) -- most planted rows must still be in the top-k; the id-by-id
            # comparison with the independent brute force below is the real check
            found = sum(int(planted[last][r] in I_start[r]) for r in range(B // 2))
            assert found >= (B // 2) * 3 // 4, f"only {found}/{B // 2} planted rows in the top-k"
        # (anisotropic dump: a planted row sits in a run of near-duplicates whose norms spread over a factor of three -- the run's
        # heavier rows out-score it; only the brute-force comparison below applies)

    # recall@k computed, not argued: the top-k of a few queries of the last batch again, by an independent fp64 brute
    # force in plain torch over this rank's shard; N > 1: the per-rank answers are all-gathered and merged on the host
    # (score desc, id asc) -- the merged answer of the timed step must equal that, id by id
    recall = None
    if args.recall_queries > 0 and not args.no_check:
        nq = min(args.recall_queries, B)
        sel = torch.cat([torch.arange(B // 2 - nq // 2, B // 2 + (nq + 1) // 2), B + torch.arange(nq)]).to(dev)
        xs = searcher.x[sel]
        try:
            ref_s, ref_i = independent_topk(shard.rows_dev_ptr(), n_local, lo, xs, k, dev)
            shard.finalize()                 # rows_dev_ptr() marks the shard dirty
            if dist is not None:
                gs = [torch.empty_like(ref_s) for _ in range(world)]
                gi = [torch.empty_like(ref_i) for _ in range(world)]
                dist.all_gather(gs, ref_s)
                dist.all_gather(gi, ref_i)
                cs, ci = torch.cat(gs, 1).cpu().numpy(), torch.cat(gi, 1).cpu().numpy()
                o = np.lexsort((ci, -cs), axis=1)[:, :k]
                ref_i = np.take_along_axis(ci, o, 1)
            else:
                ref_i = ref_i.cpu().numpy()
            got = out["I"][sel].cpu().numpy()
            recall = {f"recall_at_{kk}": float(np.mean([len(set(ref_i[r, :kk]) & set(got[r, :kk])) / kk
                                                         for r in range(ref_i.shape[0])])) for kk in (1, 5, k)}
            recall["recall_rows_checked"] = int(ref_i.shape[0])
        except (RuntimeError, TypeError, ValueError) as e:       # torch without the CUDA array interface
            recall = {"recall_error": repr(e)[:200]}
        if "recall_at_1" in recall:
            assert all(v == 1.0 for kk, v in recall.items() if kk.startswith("recall_at")), recall

    if rank == 0:
        line, kernel, alg_launch = make_line(args, world, weak, n_total, n_local, elapsed, scan_ms, scan_launches, ladder_ms,
                                             ladder_launches, stats, pairs, triggers, n_uncert)
        sm = sorted(step_ms)
        line["ms_per_step_median"] = sm[len(sm) // 2] if len(sm) % 2 else 0.5 * (sm[len(sm) // 2 - 1] + sm[len(sm) // 2])
        line["ms_per_step_min"], line["ms_per_step_max"] = sm[0], sm[-1]
        line["queries_per_sec_median_step"] = B / (line["ms_per_step_median"] / 1e3)
        line["distinct_batches"] = {"timed": min(args.steps, len(batches)), "warmup": min(max(args.warmup, 1), len(batches)),
                                    "queries": args.queries,
                                    "note": "every timed step searches a batch no earlier step has seen (BASELINE.md section 3: 1000 questions = 15 batches of 64); "
                                            "ms_per_step_* from one HIP event per step on the steps' stream (this rank), `value` from the wall clock over all steps"}
        if preflight is not None:
            line["ranks_seen"] = preflight["ranks_seen"]
            line["preflight"] = preflight
            line["per_rank"] = per_rank_ms
        if recall is not None:
            line.update(recall)
            line["recall_note"] = "id overlap with an independent fp64 brute force (plain torch) over the resident dump"
        if world == 1 and not args.no_also and B == 64 and kind == 0 and args.rows == 0:     # the default configs[1] run only
            # configs[3] / configs[4] / end-to-end, driver-timed in the same run (never part of `value`)
            also = {}
            for name, fn in (("e2e_mips_search", lambda: also_e2e(shard, args, n_total)),
                             ("exact_b512_document", lambda: also_b512_document(shard, args, n_total)),
                             ("encoder_overlap_b512", lambda: also_encoder_overlap(shard, args, dev))):
                t_leg = time.perf_counter()
                try:
                    also[name] = fn()
                except Exception as e:                      # a failing side leg must not lose the headline line
                    also[name] = {"error": repr(e)[:300]}
                also[name]["leg_seconds"] = time.perf_counter() - t_leg
            searcher = None
            shard.close()
            torch.cuda.empty_cache()
            t_leg = time.perf_counter()
            try:
                also["anisotropic_b64"] = also_anisotropic(args, dev, local)
            except Exception as e:
                also["anisotropic_b64"] = {"error": repr(e)[:300]}
            also["anisotropic_b64"]["leg_seconds"] = time.perf_counter() - t_leg
            torch.cuda.empty_cache()
            t_leg = time.perf_counter()
            try:
                also["ivf4096_b256"] = also_ivf(args, dev, local)
            except Exception as e:
                also["ivf4096_b256"] = {"error": repr(e)[:300]}
            also["ivf4096_b256"]["leg_seconds"] = time.perf_counter() - t_leg
            torch.cuda.empty_cache()
            t_leg = time.perf_counter()
            try:
                also["pq_opq96_ivf2p20_b64"] = also_pq(args, dev, local)
            except Exception as e:
                also["pq_opq96_ivf2p20_b64"] = {"error": repr(e)[:300]}
            also["pq_opq96_ivf2p20_b64"]["leg_seconds"] = time.perf_counter() - t_leg
            torch.cuda.empty_cache()
            t_leg = time.perf_counter()
            try:
                also["pq_opq96_ivf2p20_skewed_b64"] = also_pq_skewed(args, dev, local)
            except Exception as e:
                also["pq_opq96_ivf2p20_skewed_b64"] = {"error": repr(e)[:300]}
            also["pq_opq96_ivf2p20_skewed_b64"]["leg_seconds"] = time.perf_counter() - t_leg
            line["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, n_total)
        if world == 1 and not args.no_traffic:
            searcher = None
            shard.close()                         # (idempotent) the nested pass needs the HBM
            torch.cuda.empty_cache()
            t_leg = time.perf_counter()
            traffic, note = measure_traffic(args, kernel)
            line["roofline"]["traffic"] = traffic
            line["roofline"]["traffic_note"] = note + f" ({time.perf_counter() - t_leg:.0f} s)"
            if traffic is not None:
                line["roofline"]["traffic_over_algorithmic"] = traffic / alg_launch
        line["bench_wall_seconds"] = time.perf_counter() - t_start      # this process, start to the line (build, fill, legs, baseline)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
