#!/usr/bin/env python3
"""The timed CPU comparator of bench.py (`cpu_baseline`), in a process of its own.  TEST / MEASUREMENT INFRASTRUCTURE: the
product never imports this.

The FAISS-CPU IndexFlatIP execution shape (SURVEY.md 8d "CPU baseline timing") on a bounded sample, ALL host cores:
fp32 vectors resident in RAM (de-quantised once, like an index built from the dump: x = n/20 - 2, embed_utils.py:148),
the database walked in blocks, one sgemm per block against the stacked [2B,768] query rows (index.py:196-200), a running
top-k per query row (FAISS: heap_addn over the block's scores), merged over the workers at the end.

How the cores are used: numpy's OpenBLAS here is built with MAX_THREADS=64 and splits ONE [block,768]x[768,2B] product
badly over many threads (round 2: 125 GFLOP/s on 256 cores).  FAISS itself parallelises the flat search over database
blocks with OpenMP, so this does the same: OPENBLAS_NUM_THREADS=1 and one python thread per core, each running
single-threaded sgemms over its own blocks (numpy releases the GIL inside the product and inside the reductions).
The sample is sized past every cache level (default: 8 GiB of fp32 = 2.8 M rows, or a quarter of the free RAM if that
is less) and holds DISTINCT rows of the dump's distribution (int8 codes ~ 40 + 12 z, the i.i.d. dump of BASELINE
config 2).  Prints one JSON object: rows, seconds per pass, Q/s on the sample, GFLOP/s, GB/s of database bytes streamed,
threads used.
Usage: python -m oracle.cpu_baseline --batch 64 --top_k 10 [--gib 8] [--budget 12] [--threads N]"""
import os

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")        # before numpy loads its BLAS: one sgemm = one core
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import argparse                                            # noqa: E402
import json                                                # noqa: E402
import sys                                                 # noqa: E402
import time                                                # noqa: E402
from concurrent.futures import ThreadPoolExecutor          # noqa: E402

import numpy as np                                         # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.mips_oracle import flat_ip_search_fp32_resident      # noqa: E402

BLK = 8192            # rows per sgemm: 24 MiB of fp32, FAISS walks its database in blocks as well (1024 rows there)


def _free_ram_bytes() -> int:
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 16 << 30


def _make_block(b: int) -> np.ndarray:
    rng = np.random.default_rng([7, b])
    nb = rng.standard_normal((BLK, 768), dtype=np.float32)
    nb *= 12.0
    nb += 40.0
    np.rint(nb, out=nb)
    np.clip(nb, -128, 127, out=nb)
    nb /= 20.0                                             # x = n/20 - 2 (embed_utils.py:148)
    nb -= 2.0
    return nb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--top_k", type=int, default=10)
    ap.add_argument("--gib", type=float, default=8.0, help="fp32 bytes of the resident sample (capped at 1/4 of the free RAM)")
    ap.add_argument("--rows", type=int, default=0, help="explicit sample rows (overrides --gib)")
    ap.add_argument("--budget", type=float, default=12.0, help="seconds of timed passes")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    from oracle._cpus import affinity_cpus, effective_cpus, quota_cpus
    cores = affinity_cpus()
    threads = a.threads or effective_cpus()           # (the cgroup CPU quota counts: more busy threads than it allows are throttled)
    rows = a.rows or int(min(a.gib * (1 << 30), _free_ram_bytes() / 4) // (768 * 4))
    n_blocks = max(1, -(-(rows // BLK) // threads)) * threads      # the same number of blocks for every thread
    n_cpu = n_blocks * BLK
    pool = ThreadPoolExecutor(max_workers=threads)
    blocks = list(pool.map(_make_block, range(n_blocks)))
    q = np.random.default_rng(7).normal(0, 0.5, (2 * a.batch, 768)).astype(np.float32)
    # contiguous runs of blocks per worker (ids stay id_base + row), the running top-k is worker-local
    cuts = [n_blocks * w // threads for w in range(threads + 1)]

    def work(w):
        return flat_ip_search_fp32_resident(q, blocks[cuts[w]:cuts[w + 1]], a.top_k, id_base=cuts[w] * BLK)

    def one_pass():
        parts = list(pool.map(work, range(threads)))
        s = np.concatenate([p[0] for p in parts], 1)
        i = np.concatenate([p[1] for p in parts], 1)
        o = np.lexsort((i, -s), axis=1)[:, :a.top_k]       # (score desc, id asc) like the oracle
        return np.take_along_axis(s, o, 1), np.take_along_axis(i, o, 1)

    one_pass()                                             # warm-up (page faults, thread start)
    times, t_start = [], time.time()
    while len(times) < 3 or (time.time() - t_start < a.budget and len(times) < 400):
        t0 = time.time()
        D, I = one_pass()
        times.append(time.time() - t0)
    t = float(np.median(times))
    # the merged answer of the threaded passes is the single-threaded oracle's (checked on the first worker's share)
    D0, I0 = flat_ip_search_fp32_resident(q[:4], blocks, a.top_k)
    assert (I0 == I[:4]).all() or np.allclose(D0, D[:4], rtol=1e-5), "threaded CPU baseline disagrees with the oracle"
    print(json.dumps({"rows": n_cpu, "block": BLK, "seconds_per_pass": t, "passes": len(times), "cores": threads,
                      "host_cores": cores, "sample_gib": n_cpu * 768 * 4 / (1 << 30), "qps_sample": a.batch / t,
                      "gflops": 2 * (2 * a.batch) * 768 * n_cpu / t / 1e9, "db_gbytes_per_s": n_cpu * 768 * 4 / t / 1e9, "cpu_quota": quota_cpus()}))


if __name__ == "__main__":
    main()
