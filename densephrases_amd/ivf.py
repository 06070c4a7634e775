"""IVF list builder for libdph's list-major shards (BASELINE.json configs[3]).

Replaces what the reference does with FAISS at index-build time -- k-means of the coarse quantizer
(/root/reference/build_phrase_index.py:96-142, `IndexFlatIP` quantizer at :99) and `add_with_ids` into inverted
lists (:145-153) -- for the *exact in-list* variant: vectors stay int8 rows, only their order changes.

  train_centroids_resident
                    the trainer of the product path: spherical k-means (what FAISS runs for METRIC_INNER_PRODUCT indexes:
                    assignment = a search of the IndexFlatIP quantizer, centroids L2-normalised) over a sample of the
                    RESIDENT int8 rows drawn like build_phrase_index.py:60-93 (20 % x 20 % = 4 % of the rows, capped at
                    FAISS' 256 points per centroid); every iteration is one dph_kmeans_step_dev: fused MFMA assignment +
                    HIP centroid update from exact integer sums
  train_centroids   small host-side L2 Lloyd iterations over rows in host memory (tests, toy dumps)
  assign_lists      list of a row = arg-max inner product with the centroids (the quantizer is an IndexFlatIP);
                    ``assign_lists_gpu`` does it with the same kernel and re-checks near-ties in float64
  build_list_major  permutation of the rows into contiguous lists, each padded to a multiple of 32 rows (one scan
                    tile never straddles two lists); returns the stored rows, row_ids (-1 = padding) and tile_list
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

TILE_ROWS = 32


def dequant(rows: np.ndarray, offset: float = -2.0, scale: float = 20.0) -> np.ndarray:
    return rows.astype(np.float32) / np.float32(scale) + np.float32(offset)


def _assign_dev(x, c, bias=None, chunk: int = 1 << 22):
    """arg-max_l <x_r, c_l> (+ bias_l) for GPU tensors x [n,768], c [nlist,768] through libdph (MFMA f32 GEMM fused with
    the arg-max: no score matrix); returns (best int64 [n], gap fp32 [n])."""
    import ctypes as C
    import torch
    from . import _lib
    n, nlist = x.shape[0], c.shape[0]
    best = torch.empty(n, dtype=torch.int32, device=x.device)
    gap = torch.empty(n, dtype=torch.float32, device=x.device)
    st = torch.cuda.current_stream(x.device).cuda_stream
    vp = C.c_void_p
    for r0 in range(0, n, chunk):
        m = min(chunk, n - r0)
        _lib._chk(_lib.lib.dph_ivf_assign_dev(x.device.index, vp(x[r0:].data_ptr()), m, vp(c.data_ptr()), nlist,
                                              vp(bias.data_ptr()) if bias is not None else None, None,
                                              vp(best[r0:].data_ptr()), vp(gap[r0:].data_ptr()), vp(st)))
    return best.to(torch.int64), gap


def train_centroids(rows_int8: np.ndarray, nlist: int, iters: int = 10, seed: int = 0, offset: float = -2.0,
                    scale: float = 20.0) -> np.ndarray:
    import torch
    on_gpu = torch.cuda.is_available()
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    if rows_int8.shape[0] < nlist:
        raise ValueError(f"train_centroids: {rows_int8.shape[0]} training rows for {nlist} lists (need at least one row per list)")
    x = torch.from_numpy(dequant(rows_int8, offset, scale)).to(dev).contiguous()
    g = torch.Generator(device="cpu").manual_seed(seed)
    c = x[torch.randperm(x.shape[0], generator=g)[:nlist].to(dev)].clone()
    for _ in range(iters):
        # L2 Lloyd step (what faiss.Clustering does under the IVF trainer): argmin ||x-c||^2 = argmax <x,c> - ||c||^2/2
        if on_gpu:
            a, _ = _assign_dev(x, c.contiguous(), bias=(-0.5 * (c * c).sum(1)).contiguous())
        else:
            d2 = (x * x).sum(1, keepdim=True) - 2.0 * (x @ c.T) + (c * c).sum(1)[None, :]
            a = d2.argmin(1)
        sums = torch.zeros_like(c).index_add_(0, a, x)
        cnt = torch.bincount(a, minlength=nlist).to(x.dtype).clamp_min(1.0)
        newc = sums / cnt[:, None]
        empty = torch.bincount(a, minlength=nlist) == 0
        newc[empty] = c[empty]
        c = newc
    return c.cpu().numpy().astype(np.float32)


def assign_lists(rows_int8: np.ndarray, centroids: np.ndarray, offset: float = -2.0, scale: float = 20.0,
                 block: int = 1 << 16) -> np.ndarray:
    """arg-max <x, centroid> in float64 (ties to the lowest list id), block-wise."""
    out = np.empty(rows_int8.shape[0], dtype=np.int32)
    c64 = centroids.astype(np.float64)
    for b0 in range(0, rows_int8.shape[0], block):
        x = dequant(rows_int8[b0:b0 + block], offset, scale).astype(np.float64)
        out[b0:b0 + block] = np.argmax(x @ c64.T, axis=1)
    return out


def assign_lists_gpu(rows_int8: np.ndarray, centroids: np.ndarray, offset: float = -2.0, scale: float = 20.0,
                     block: int = 1 << 18) -> np.ndarray:
    """``assign_lists`` on the GPU: libdph's MFMA GEMM + arg-max per block of rows; rows whose best two fp32 scores are
    closer than the fp32 error band are re-assigned in float64 on the host, so the result equals ``assign_lists``."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    c = torch.from_numpy(np.ascontiguousarray(centroids, dtype=np.float32)).to(dev)
    cmax = float(np.sqrt((centroids.astype(np.float64) ** 2).sum(1).max()))
    out = np.empty(rows_int8.shape[0], dtype=np.int32)
    c64 = centroids.astype(np.float64)
    for b0 in range(0, rows_int8.shape[0], block):
        x = torch.from_numpy(dequant(rows_int8[b0:b0 + block], offset, scale)).to(dev).contiguous()
        best, gap = _assign_dev(x, c)
        best, gap = best.cpu().numpy(), gap.cpu().numpy()
        band = 4.0 * 1.5 * 768.0 * 5.97e-8 * np.sqrt((x * x).sum(1).cpu().numpy().astype(np.float64)) * cmax
        near = np.nonzero(gap.astype(np.float64) <= band)[0]
        if near.size:
            xs = dequant(rows_int8[b0 + near], offset, scale).astype(np.float64)
            best[near] = np.argmax(xs @ c64.T, axis=1)
        out[b0:b0 + block] = best
    return out


def _lut_dev(offset: float, scale: float, dev):
    """the reference's fp32 value of every int8 code (two roundings: fl(fl(n)/scale) + offset, embed_utils.py:148-149)
    as a device table.  A torch expression would not do: ``t / 20.0`` on the GPU is ``t * fl(1/20)``, one ulp off for some
    codes -- enough to flip a 1e-10 near-tie between two lists."""
    import torch
    codes = np.arange(-128, 128, dtype=np.int8)
    return torch.from_numpy(dequant(codes, offset, scale)).to(dev)


def _dequant_dev(rows_i8, lut):
    import torch
    return lut.index_select(0, rows_i8.reshape(-1).to(torch.int32) + 128).reshape(rows_i8.shape)


def assign_lists_resident(shard, centroids: np.ndarray, offset: float = -2.0, scale: float = 20.0, block: int = 1 << 22):
    """``assign_lists`` for the rows of a RESIDENT shard, where they lie in HBM (dph_index_assign_dev: int8 rows
    de-quantised through the shard's LUT inside the fused MFMA GEMM + arg-max -- 170 M rows x 4096 lists never leave the
    GPU and no score matrix exists).  Near-ties (gap inside the fp32 error band) are re-assigned in float64 on the device.
    Returns an int32 torch tensor [n_rows] on the shard's GPU."""
    import torch
    dev = torch.device("cuda", shard.device)
    n, nlist = shard.n_rows, centroids.shape[0]
    c = torch.from_numpy(np.ascontiguousarray(centroids, dtype=np.float32)).to(dev)
    c64 = c.to(torch.float64)
    lut = _lut_dev(offset, scale, dev)
    cmax = float(np.sqrt((centroids.astype(np.float64) ** 2).sum(1).max()))
    best = torch.empty(n, dtype=torch.int32, device=dev)
    gap = torch.empty(n, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    class _Rows:
        def __init__(self, ptr):
            self.__cuda_array_interface__ = {"shape": (n, 768), "typestr": "|i1", "data": (int(ptr), False), "version": 2}
    finalized_ptr = shard.rows_dev_ptr()                 # (marks the shard dirty: the caller finalizes afterwards anyway)
    rows = torch.as_tensor(_Rows(finalized_ptr), device=dev)
    for r0 in range(0, n, block):
        m = min(block, n - r0)
        shard.assign_lists_dev(c.data_ptr(), nlist, best[r0:].data_ptr(), gap[r0:].data_ptr(), row0=r0, n=m, stream=st)
        # |fp32 MFMA dot - exact| <= 768 * 2^-24 * ||x|| * max||c||, with the slack of dph_coarse_select_kernel
        for s0 in range(r0, r0 + m, 1 << 20):
            s1 = min(s0 + (1 << 20), r0 + m)
            xn = _dequant_dev(rows[s0:s1], lut).square_().sum(1).sqrt_()
            near = torch.nonzero(gap[s0:s1] <= 4.0 * 1.5 * 768.0 * 5.97e-8 * cmax * xn).flatten()
            del xn
            if near.numel():
                xs = _dequant_dev(rows[s0 + near], lut).to(torch.float64)
                best[s0 + near] = torch.argmax(xs @ c64.T, dim=1).to(torch.int32)
    return best


SAMPLE_RATIO = 0.2 * 0.2            # build_phrase_index.py:60-93: 20 % of the documents x 20 % of their vectors
MAX_POINTS_PER_CENTROID = 256       # faiss ClusteringParameters defaults: more training points are sub-sampled away,
MIN_POINTS_PER_CENTROID = 39        # fewer draw a warning


def kmeans_sample_size(n_rows: int, nlist: int, train_rows: Optional[int] = None) -> int:
    """rows the trainer looks at: 4 % of the dump like the reference's sample_data, at least FAISS' 39 and at most its 256
    points per centroid, never more than there are"""
    m = int(train_rows) if train_rows else int(n_rows * SAMPLE_RATIO)
    m = max(m, MIN_POINTS_PER_CENTROID * nlist)
    m = min(m, MAX_POINTS_PER_CENTROID * nlist, n_rows)
    if m < nlist:
        raise ValueError(f"k-means: {m} training rows for {nlist} lists (a shard needs at least one row per list)")
    return m


def split_empty_lists(c, counts, eps: float = 1.0 / 1024.0):
    """FAISS' answer to empty clusters (Clustering.cpp split_clusters): an empty list takes the centroid of a large one,
    the two copies perturbed symmetrically by +-eps on alternating components.  Donors: the largest lists, in order
    (FAISS draws them at random in proportion to their size).  c [nlist,768] and counts [nlist] are torch tensors on one
    device; c is modified in place.  Returns the number of lists split."""
    import torch
    empty = torch.nonzero(counts == 0).flatten()
    ne = int(empty.numel())
    if ne == 0:
        return 0
    donors = torch.argsort(counts, descending=True)[:ne]
    donors = donors[counts[donors] > 1]
    empty = empty[:donors.numel()]
    sign = torch.ones(c.shape[1], dtype=c.dtype, device=c.device)
    sign[1::2] = -1.0
    base = c[donors].clone()
    c[empty] = base * (1.0 + eps * sign)
    c[donors] = base * (1.0 - eps * sign)
    return int(empty.numel())


def train_centroids_resident(shard, nlist: int, iters: int = 10, train_rows: Optional[int] = None, seed: int = 0,
                             spherical: bool = True, offset: float = -2.0, scale: float = 20.0, return_info: bool = False):
    """k-means of the coarse quantizer over a sample of the rows of a RESIDENT flat shard, on the GPU end to end (module
    docstring).  Returns centroids fp32 [nlist,768] (numpy) and, with ``return_info``, a dict with the sample size, the
    per-iteration seconds and the list sizes of the last iteration."""
    import time
    import torch
    dev = torch.device("cuda", shard.device)
    n = shard.ntotal
    m = kmeans_sample_size(n, nlist, train_rows)
    g = torch.Generator(device=dev).manual_seed(seed)
    # sample without replacement; ascending row order keeps the gather's reads going one way through HBM
    if m == n:
        pick = torch.arange(n, dtype=torch.int64, device=dev)
    elif n <= (1 << 28):
        pick = torch.randperm(n, generator=g, device=dev)[:m].sort().values
    else:                                # a permutation of > 2^28 entries is not worth its 2 GB: rejection-free stride + jitter
        step = n / m
        pick = (torch.arange(m, dtype=torch.float64, device=dev) * step
                + torch.rand(m, generator=g, device=dev, dtype=torch.float64) * step).to(torch.int64).clamp_(max=n - 1)
    st = torch.cuda.current_stream(dev).cuda_stream
    sample = torch.empty((m, 768), dtype=torch.int8, device=dev)
    shard.gather_rows_dev(pick.data_ptr(), m, sample.data_ptr(), stream=st)
    # initial centroids: distinct random training points (FAISS: a random subset of the training set)
    lut = _lut_dev(offset, scale, dev)
    first = torch.randperm(m, generator=g, device=dev)[:nlist]
    c = _dequant_dev(sample[first], lut).contiguous()
    if spherical:
        c = c / c.norm(dim=1, keepdim=True).clamp_min(1e-20)
    assign = torch.empty(m, dtype=torch.int32, device=dev)
    gap = torch.empty(m, dtype=torch.float32, device=dev)
    counts = torch.empty(nlist, dtype=torch.int32, device=dev)
    secs, splits = [], []
    for _ in range(iters):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        bias = None if spherical else (-0.5 * (c * c).sum(1)).contiguous()
        shard.kmeans_step_dev(sample.data_ptr(), m, c.data_ptr(), nlist, assign.data_ptr(), gap.data_ptr(), counts.data_ptr(),
                              bias_ptr=bias.data_ptr() if bias is not None else 0, spherical=spherical, stream=st)
        splits.append(split_empty_lists(c, counts))
        torch.cuda.synchronize(dev)
        secs.append(time.perf_counter() - t0)
    out = c.cpu().numpy().astype(np.float32)
    if return_info:
        cc = counts.cpu().numpy()
        return out, {"sample_rows": m, "iters": iters, "seconds_per_iter": secs, "lists_split_per_iter": splits,
                     "largest_list_in_sample": int(cc.max()), "empty_lists_last_iter": int((cc == 0).sum())}
    return out


def make_list_major_resident(shard, nlist: int, centroids: Optional[np.ndarray] = None, iters: int = 10,
                             train_rows: Optional[int] = None, seed: int = 0, offset: float = -2.0, scale: float = 20.0,
                             rehome: bool = False):
    """A FLAT shard whose rows are resident in HBM -> list-major IVF shard, on the GPU end to end: centroids (given, or
    ``train_centroids_resident``), ``assign_lists_resident``, then libdph's device-side list builder (radix sort by
    (list, id) + row gather, dph_index_make_list_major); ``rehome=True`` additionally moves the permuted rows into a fresh
    allocation (dph_index_rehome_rows: a compaction utility that needs a second copy of the rows in HBM and was measured to
    bring no speed-up -- off by default, and a failure of it leaves the valid shard as it is).  Returns (centroids fp32 [nlist,768], assign int32 torch tensor [n]).  The caller sets
    idx2id / f2o (before or after) and finalizes."""
    import torch
    dev = torch.device("cuda", shard.device)
    if centroids is None:
        centroids = train_centroids_resident(shard, nlist, iters=iters, train_rows=train_rows, seed=seed, offset=offset, scale=scale)
    centroids = np.ascontiguousarray(centroids, dtype=np.float32)
    assign = assign_lists_resident(shard, centroids, offset=offset, scale=scale)
    shard.make_list_major(assign.data_ptr(), centroids, stream=torch.cuda.current_stream(dev).cuda_stream)
    if rehome:
        # the permuted copy was allocated while the original still filled half of the HBM; now that the original is gone,
        # move it into a fresh allocation (dph.h: dph_index_rehome_rows -- large physical fragments again)
        torch.cuda.empty_cache()
        try:
            shard.rehome_rows(stream=torch.cuda.current_stream(dev).cuda_stream)
        except Exception as e:                       # DPH_E_NOMEM right after the builder freed the original: the shard is valid as built
            import logging
            logging.getLogger(__name__).warning("rehome of the list-major rows skipped: %s", e)
    return centroids, assign


def build_list_major(rows_int8: np.ndarray, assign: np.ndarray, nlist: int,
                     id_base: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(stored_rows int8 [n_store,768], row_ids int64 [n_store], tile_list int32 [n_store/32]).
    Inside a list rows keep their id order (so ties still resolve to the lowest id first)."""
    n = rows_int8.shape[0]
    order = np.argsort(assign, kind="stable")
    counts = np.bincount(assign, minlength=nlist)
    padded = (counts + TILE_ROWS - 1) // TILE_ROWS * TILE_ROWS
    starts = np.concatenate([[0], np.cumsum(padded)])
    n_store = int(starts[-1])
    stored = np.zeros((n_store, rows_int8.shape[1]), dtype=np.int8)
    row_ids = np.full(n_store, -1, dtype=np.int64)
    src0 = np.concatenate([[0], np.cumsum(counts)])
    for l in range(nlist):
        k = int(counts[l])
        if k:
            sel = order[src0[l]:src0[l] + k]
            stored[starts[l]:starts[l] + k] = rows_int8[sel]
            row_ids[starts[l]:starts[l] + k] = sel.astype(np.int64) + id_base
    tile_list = np.repeat(np.arange(nlist, dtype=np.int32), (padded // TILE_ROWS).astype(np.int64))
    assert tile_list.shape[0] * TILE_ROWS == n_store and (row_ids >= 0).sum() == n
    return stored, row_ids, tile_list
