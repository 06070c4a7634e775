"""Host replica of libdph's on-device synthetic dump generator (dph_index_fill_synthetic, csrc/dph_scan.hip
``dph_fill_kernel``): BASELINE.md config 2 -- rows i.i.d. ~ float_to_int8(N(0, 0.6^2), -2, 20) = 40 + 12 z,
generated with integer arithmetic only (Irwin-Hall sum of four hashed bytes) so the GPU and this numpy code
agree bit for bit.  Used by the tests and by bench.py's bounded CPU sample."""
from __future__ import annotations

import numpy as np

DIM = 768


def _hash32(lo: np.ndarray, hi: np.ndarray, seed: int) -> np.ndarray:
    m = np.uint64(0xFFFFFFFF)
    h = ((lo * np.uint64(0x9E3779B1)) & m) ^ ((hi * np.uint64(0x85EBCA77) + np.uint64(seed & 0xFFFFFFFF)) & m)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & m
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & m
    h ^= h >> np.uint64(16)
    return h


def synthetic_rows(row0: int, n: int, seed: int = 42) -> np.ndarray:
    """int8 [n, 768]: rows row0 .. row0+n-1 of the synthetic dump (global row index = id_base + local row)."""
    e = (np.arange(n * DIM, dtype=np.uint64) + np.uint64(row0 * DIM))
    lo = e & np.uint64(0xFFFFFFFF)
    hi = (e >> np.uint64(32)) ^ np.uint64((seed >> 32) & 0xFFFFFFFF)
    h = _hash32(lo, hi, seed)
    s = ((h & np.uint64(255)) + ((h >> np.uint64(8)) & np.uint64(255)) + ((h >> np.uint64(16)) & np.uint64(255))
         + (h >> np.uint64(24))).astype(np.int64)
    v = 40 + (((s - 510) * 5321 + 32768) >> 16)
    return np.clip(v, -128, 127).astype(np.int8).reshape(n, DIM)


def synthetic_queries(n: int, seed: int = 1234, planted_rows=None, noise: float = 0.1) -> np.ndarray:
    """fp32 [n, 768] query rows.  With ``planted_rows`` (int8 [n,768]) every query is a de-quantised stored row
    plus N(0, noise^2): the nearest neighbour is known."""
    rng = np.random.default_rng(seed)
    if planted_rows is not None:
        base = planted_rows.astype(np.float32) / 20.0 - 2.0
        return (base + rng.normal(0.0, noise, base.shape)).astype(np.float32)
    return rng.normal(0.0, 0.6, (n, DIM)).astype(np.float32)
