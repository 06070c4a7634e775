#!/usr/bin/env python3
"""The timed CPU comparator of bench.py's PQ leg (`also.pq_opq96_ivf2p20_b64.cpu_baseline`), in a process of its own.  TEST /
MEASUREMENT INFRASTRUCTURE: the product never imports this.

What FAISS-CPU runs for the reference's released index (index.py:30-33 read_index, :53 nprobe = 256, :200 search): IndexPreTransform
(OPQ) -> IndexIVFPQ with an IndexFlatIP coarse quantizer, inner product, by_residual.  Restated with numpy on the host cores over the
SAME synthetic index the GPU leg searches (densephrases_amd.synth.synthetic_pq_parts: the index is a function of its seed, so the 16 GB
of codes are never materialised -- a probed list's codes are cut out of the 2^20-code block they are rolled copies of):

    x'      = x A^T                                            one sgemm
    coarse  = top-nprobe of x' C^T over ALL nlist centroids    blocked sgemm (one thread per block of centroids) + argpartition per row
    LUT     = <x'_m, codeword[m][j]>                           einsum
    ADC     = dis0 + sum_m LUT[m][code[m]] over the probed lists, k best per row   (one thread per query row)

Arithmetic and tie order are the oracle's (oracle/ivfpq_oracle.py: float64 accumulation rounded once for x', coarse scores and LUT;
fp32 sequential sum in m; (score desc, id asc)) -- the first rows are checked against it.  A port (FAISS has SIMD scan kernels and
precomputed tables this does not), labelled so.  Prints one JSON object.
Usage: python -m oracle.cpu_baseline_pq --batch 64 --nlist 1048576 --codes 170000000 [--budget 20]"""
import os

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")

import argparse                                            # noqa: E402
import json                                                # noqa: E402
import sys                                                 # noqa: E402
import time                                                # noqa: E402
from concurrent.futures import ThreadPoolExecutor          # noqa: E402

import numpy as np                                         # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def codes_of(block, pos):
    """codes of index positions `pos` (int64): block[(p % 2^20 - (p >> 20) % 97) mod 2^20]"""
    n = block.shape[0]
    return block[(pos % n - (pos // n) % 97) % n]


def search_row(xp_row, probe, dis0, list_off, pqc, block, k):
    """ADC over the probed lists of one query row: (D [k], I [k]), the oracle's arithmetic"""
    M, ksub, dsub = pqc.shape
    table = np.einsum("mt,mjt->mj", xp_row.reshape(M, dsub).astype(np.float64), pqc.astype(np.float64)).astype(np.float32)
    lens = (list_off[probe + 1] - list_off[probe]).astype(np.int64)
    tot = int(lens.sum())
    if tot == 0:
        return np.full(k, -3.4028235e38, np.float32), np.full(k, -1, np.int64)
    # positions of all probed codes, and the coarse score of the list each belongs to
    starts = np.repeat(list_off[probe], lens)
    within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
    pos = starts + within
    acc = np.repeat(dis0, lens).astype(np.float32)
    codes = codes_of(block, pos)
    for m in range(M):
        acc = (acc + table[m][codes[:, m]]).astype(np.float32)           # sequential in m like FAISS' scalar scan
    kk = min(k, tot)
    part = np.argpartition(-acc, kk - 1)[:kk]
    o = part[np.lexsort((pos[part], -acc[part].astype(np.float64)))]
    D = np.full(k, -3.4028235e38, np.float32)
    I = np.full(k, -1, np.int64)
    D[:kk], I[:kk] = acc[o], pos[o]
    return D, I


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--top_k", type=int, default=10)
    ap.add_argument("--nlist", type=int, default=1 << 20)
    ap.add_argument("--codes", type=int, default=170_000_000)
    ap.add_argument("--nprobe", type=int, default=256)
    ap.add_argument("--M", type=int, default=96)
    ap.add_argument("--budget", type=float, default=20.0)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    from densephrases_amd.synth import synthetic_pq_parts
    from oracle._cpus import affinity_cpus, effective_cpus
    cores = affinity_cpus()
    threads = a.threads or effective_cpus()           # (the cgroup CPU quota counts: more busy threads than it allows are throttled)
    t0 = time.time()
    sizes, A, cent, pqc, block = synthetic_pq_parts(a.codes, a.nlist, a.M)
    list_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    build_s = time.time() - t0
    R = 2 * a.batch
    x = np.random.default_rng(3).normal(0, 0.5, (R, 768)).astype(np.float32)         # bench.py also_pq's queries
    pool = ThreadPoolExecutor(max_workers=threads)
    nprobe = min(a.nprobe, a.nlist)
    cblk = 8192
    cuts = list(range(0, a.nlist, cblk))

    def coarse_block(c0):
        # float64 accumulation like the oracle would be 10x the work of FAISS' sgemm: the timed coarse step is the fp32 product FAISS
        # runs; the probe set of the checked rows is recomputed in float64 below
        return xp @ cent[c0:c0 + cblk].T

    def one_batch():
        nonlocal xp
        xp = (x.astype(np.float64) @ A.astype(np.float64).T).astype(np.float32)
        s = np.concatenate(list(pool.map(coarse_block, cuts)), axis=1)              # [R, nlist] fp32
        def row(r):
            pr = np.argpartition(-s[r], nprobe - 1)[:nprobe]
            pr = pr[np.lexsort((pr, -s[r][pr].astype(np.float64)))]
            return search_row(xp[r], pr, s[r][pr], list_off, pqc, block, a.top_k)
        out = list(pool.map(row, range(R)))
        return np.stack([o[0] for o in out]), np.stack([o[1] for o in out]), s

    xp = None
    one_batch()
    times, t_start = [], time.time()
    while len(times) < 2 or (time.time() - t_start < a.budget and len(times) < 100):
        t1 = time.time()
        D, I, s = one_batch()
        times.append(time.time() - t1)
    t = float(np.median(times))
    # the first rows again with the oracle's float64 coarse scores: same answer outside fp32 near-ties of the probe boundary
    from oracle import ivfpq_oracle as P
    agree = 0
    for r in range(2):
        lists, dis0 = P.coarse_probe(xp[r:r + 1], cent, nprobe)
        Dr, Ir = search_row(xp[r], lists[0], dis0[0], list_off, pqc, block, a.top_k)
        agree += int(len(set(Ir.tolist()) & set(I[r].tolist())) >= a.top_k - 1)
    assert agree == 2, "threaded CPU IVFPQ baseline disagrees with the float64 probe set"
    lens = sizes[np.argpartition(-s, nprobe - 1, axis=1)[:, :nprobe]]
    print(json.dumps({"batch": a.batch, "rows": R, "nlist": a.nlist, "codes": a.codes, "nprobe": nprobe, "M": a.M, "seconds_per_batch": t,
                      "batches": len(times), "cores": threads, "host_cores": cores, "qps": a.batch / t, "index_build_seconds": build_s,
                      "coarse_gflops": 2.0 * R * 768 * a.nlist / 1e9, "codes_scored_per_batch": float(lens.sum())}))


if __name__ == "__main__":
    main()
