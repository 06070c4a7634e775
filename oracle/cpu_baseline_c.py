#!/usr/bin/env python3
"""bench.py's `cpu_baseline`, third implementation: the flat search in plain C with an AVX-512 sgemm micro-kernel and OpenMP
(oracle/csrc/cpu_flat_avx512.c, compiled by __graft_entry__.build() into oracle/_cbuild/libcpuflat.so) -- FAISS-CPU IndexFlatIP's
execution shape (index.py:200: blocks of the fp32 database per thread, an sgemm kernel, a running top-k, a merge) with a kernel that
uses what the host has.  TEST / MEASUREMENT INFRASTRUCTURE: the product never imports this.  Prints one JSON object like the other
two comparators (oracle/cpu_baseline.py: numpy / OpenBLAS, oracle/cpu_baseline_torch.py: torch.mm / MKL); bench.py reports the fastest
of the three as `cpu_baseline.value`.
Usage: python -m oracle.cpu_baseline_c --batch 64 --top_k 10 [--gib 8] [--budget 12]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "oracle", "csrc", "cpu_flat_avx512.c")
OUT = os.path.join(ROOT, "oracle", "_cbuild", "libcpuflat.so")


def build(force: bool = False) -> str:
    """gcc -O3 -mavx512f -fopenmp (the GPU box only uses the prebuilt file: it travels with the snapshot)"""
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        tmp = OUT + f".tmp{os.getpid()}"
        subprocess.run([os.environ.get("CC", "gcc"), "-O3", "-mavx512f", "-mfma", "-fopenmp", "-shared", "-fPIC", SRC, "-o", tmp], check=True)
        os.replace(tmp, OUT)
    return OUT


def load():
    lib = C.CDLL(build())
    lib.cpu_flat_ip_topk.restype = C.c_int
    lib.cpu_flat_ip_topk.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    return lib


def search(lib, db: np.ndarray, q: np.ndarray, k: int, threads: int = 0, id_base: int = 0):
    """[nq,k] scores (desc) and ids of the k largest <q, row> over db [N,768] fp32 (nq padded to a multiple of 16 inside)"""
    nq = q.shape[0]
    nq16 = (nq + 15) // 16 * 16
    qt = np.zeros((768, nq16), np.float32)
    qt[:, :nq] = q.T
    qt = np.ascontiguousarray(qt)
    D = np.empty((nq16, k), np.float32)
    I = np.empty((nq16, k), np.int64)
    rc = lib.cpu_flat_ip_topk(db.ctypes.data, db.shape[0], id_base, qt.ctypes.data, nq16, k, D.ctypes.data, I.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError(f"cpu_flat_ip_topk: {rc}")
    return D[:nq], I[:nq]


def _free_ram_bytes() -> int:
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 16 << 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--top_k", type=int, default=10)
    ap.add_argument("--gib", type=float, default=8.0)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--budget", type=float, default=12.0)
    a = ap.parse_args()
    lib = load()
    if not lib.cpu_flat_has_avx512():
        raise SystemExit("this host has no AVX-512")
    from oracle._cpus import affinity_cpus, effective_cpus, quota_cpus
    cores, usable = affinity_cpus(), effective_cpus()
    rows = a.rows or int(min(a.gib * (1 << 30), _free_ram_bytes() / 4) // (768 * 4))
    nb = min(rows, 65536)
    rows = max(1, rows // nb) * nb
    rng = np.random.default_rng(7)
    # the dump's distribution (x = n/20 - 2, n = clip(rint(40 + 12 z))): one block drawn, the others that block with its columns rolled
    base = (np.clip(np.rint(rng.standard_normal((nb, 768), dtype=np.float32) * 12.0 + 40.0), -128, 127) / 20.0 - 2.0).astype(np.float32)
    db = np.empty((rows, 768), np.float32)
    for i, r0 in enumerate(range(0, rows, nb)):
        db[r0:r0 + nb] = base if i == 0 else np.roll(base, i % 768, axis=1)
    q = rng.normal(0, 0.5, (2 * a.batch, 768)).astype(np.float32)
    per = {}
    best_d = None
    for threads in sorted({usable, max(1, usable // 2)} if usable < cores else {cores, max(1, cores // 2)}, reverse=True):
        search(lib, db[: 3 * 4096], q, a.top_k, threads)                      # warm-up (thread team)
        times, t_start = [], time.time()
        while len(times) < 3 or (time.time() - t_start < a.budget / 2 and len(times) < 400):
            t0 = time.time()
            D, I = search(lib, db, q, a.top_k, threads)
            times.append(time.time() - t0)
        t = float(np.median(times))
        per[threads] = {"seconds_per_pass": t, "passes": len(times), "gflops": 2 * (2 * a.batch) * 768 * rows / t / 1e9}
        best_d = (D, I)
    # the answer is the oracle's (a few rows over a prefix of the database, against the numpy restatement)
    from oracle.mips_oracle import flat_ip_search_fp32_resident
    sub = db[: min(rows, 200_000)]
    D0, I0 = flat_ip_search_fp32_resident(q[:4], [sub], a.top_k)
    D1, I1 = search(lib, sub, q[:4], a.top_k, 0)
    assert (I0 == I1).all() or np.allclose(D0, D1, rtol=1e-5), "C CPU baseline disagrees with the oracle"
    threads = min(per, key=lambda th: per[th]["seconds_per_pass"])
    t = per[threads]["seconds_per_pass"]
    print(json.dumps({"rows": rows, "block": 3, "seconds_per_pass": t, "passes": per[threads]["passes"], "cores": threads, "host_cores": cores,
                      "sample_gib": rows * 768 * 4 / (1 << 30), "qps_sample": a.batch / t, "gflops": per[threads]["gflops"],
                      "db_gbytes_per_s": rows * 768 * 4 / t / 1e9, "per_threads": {str(k): v for k, v in per.items()}, "cpu_quota": quota_cpus()}))


if __name__ == "__main__":
    main()
