"""ctypes binding of libdph (include/dph.h).  The library is built in-tree by ``densephrases_amd.build`` and
loaded from ``densephrases_amd/csrc/libdph.so``; there is NO python/CPU fallback: if the shared object is
missing or fails to load, importing this module raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DPH_LIBRARY: another BUILD of libdph (tools/scan_diag.py's timing variants); never anything but a libdph
LIB_PATH = os.environ.get("DPH_LIBRARY") or os.path.join(_HERE, "csrc", "libdph.so")
if os.environ.get("DPH_LIBRARY"):
    import sys as _sys
    print(f"[densephrases_amd] DPH_LIBRARY overrides the product library: loading {LIB_PATH}", file=_sys.stderr)

DIM = 768
DPH_E_UNCERTIFIED = -6


class DphError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libdph error {code}: {msg}")
        self.code = code


class SearchStats(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("rows", "certified_fast", "certified_wide", "exact_fallback",
                                         "uncertified", "scan_launches", "fused_stride", "certified_reselect", "nonfinite")]


ROW_OK, ROW_UNCERTIFIED, ROW_DEFERRED, ROW_NONFINITE = 0, 1, 2, 3      # include/dph.h DPH_ROW_*


def _load() -> C.CDLL:
    # torch ships its own libamdhip64; whichever HIP runtime is loaded first owns the process.  Load torch's first so
    # that torch tensors (device memory, streams, torch.distributed) and libdph always share one runtime -- loading
    # libdph first made a later `import torch` report "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except Exception:          # torch is plumbing, not a hard dependency of the C ABI
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(hipcc --offload-arch=gfx950); densephrases_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i64, i32, f32p = C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_float)
    sig = {
        "dph_abi_version": (C.c_int, []),
        "dph_last_error": (C.c_char_p, []),
        "dph_device_count": (C.c_int, []),
        "dph_index_create": (C.c_int, [i32, i64, i64, C.POINTER(vp)]),
        "dph_index_destroy": (C.c_int, [vp]),
        "dph_index_set_codec": (C.c_int, [vp, C.c_float, C.c_float]),
        "dph_index_upload_rows": (C.c_int, [vp, i64, i64, vp]),
        "dph_index_upload_rows_async": (C.c_int, [vp, i64, i64, vp, vp]),
        "dph_host_alloc_pinned": (C.c_int, [C.c_size_t, C.POINTER(vp)]),
        "dph_host_free_pinned": (C.c_int, [vp]),
        "dph_stream_synchronize": (C.c_int, [i32, vp]),
        "dph_index_fill_synthetic": (C.c_int, [vp, C.c_uint64, vp]),
        "dph_index_fill_synthetic_kind": (C.c_int, [vp, C.c_uint64, i32, vp]),
        "dph_index_shard_stats": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
        "dph_index_set_tuning": (C.c_int, [vp, C.c_char_p, vp, i32]),
        "dph_scan_counters": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64)]),
        "dph_index_rehome_rows": (C.c_int, [vp, vp]),
        "dph_index_gather_rows_dev": (C.c_int, [vp, vp, i64, vp, vp]),
        "dph_kmeans_step_dev": (C.c_int, [vp, vp, i64, vp, i32, vp, i32, vp, vp, vp, vp]),
        "dph_index_stored_rows": (i64, [vp]),
        "dph_debug_wave_pairs": (C.c_int, [vp, i32, vp, i32, C.POINTER(C.c_int)]),
        "dph_index_set_idx2id": (C.c_int, [vp, vp, vp]),
        "dph_index_set_f2o": (C.c_int, [vp, i64, vp, vp, vp]),
        "dph_index_set_id_groups": (C.c_int, [vp, i32, vp, vp]),
        "dph_index_finalize": (C.c_int, [vp, vp]),
        "dph_index_ntotal": (i64, [vp]),
        "dph_index_dim": (C.c_int, [vp]),
        "dph_index_device": (C.c_int, [vp]),
        "dph_index_rows_dev": (vp, [vp]),
        "dph_search": (C.c_int, [vp, vp, i64, i32, vp, vp]),
        "dph_search_dev": (C.c_int, [vp, vp, i64, i32, vp, vp, vp, vp]),
        "dph_search_sample_dev": (C.c_int, [vp, vp, i64, vp, vp]),
        "dph_union_bounds_dev": (C.c_int, [i32, vp, i32, i64, vp, vp]),
        "dph_search_bounded_dev": (C.c_int, [vp, vp, i64, i32, vp, vp, vp, vp, vp, vp]),
        "dph_search_get_stats": (C.c_int, [vp, C.POINTER(SearchStats)]),
        "dph_index_set_row_ids": (C.c_int, [vp, vp, i64]),
        "dph_index_set_ivf": (C.c_int, [vp, i32, vp, vp]),
        "dph_search_ivf": (C.c_int, [vp, vp, i64, i32, i32, vp, vp]),
        "dph_search_ivf_dev": (C.c_int, [vp, vp, i64, i32, i32, vp, vp, vp, vp]),
        "dph_ivf_assign_dev": (C.c_int, [i32, vp, i64, vp, i32, vp, vp, vp, vp, vp]),
        "dph_index_assign_dev": (C.c_int, [vp, i64, i64, vp, i32, vp, vp, vp, vp]),
        "dph_index_make_list_major": (C.c_int, [vp, vp, i32, vp, vp]),
        "dph_reconstruct": (C.c_int, [vp, i64, vp]),
        "dph_id2docword": (C.c_int, [vp, vp, i64, vp, vp]),
        "dph_rescore": (C.c_int, [vp, i32, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
        "dph_rescore_dev": (C.c_int, [vp, i32, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "dph_score_vecs_dev": (C.c_int, [i32, vp, vp, i64, i64, vp, vp]),
        "dph_score_vecs_bwd_dev": (C.c_int, [i32, vp, vp, i64, i64, vp, vp]),
        "dph_dense_logits_dev": (C.c_int, [i32, vp, vp, i64, i64, vp, vp]),
        "dph_merge_topk_dev": (C.c_int, [i32, vp, vp, i32, i64, i64, i32, vp, vp, vp, vp]),
        "dph_merge_records_dev": (C.c_int, [i32, vp, vp, vp, vp, vp, vp, i32, i64, i64, i32, vp, vp, vp, vp, vp, vp]),
        "dph_profile_enable": (C.c_int, [vp, i32]),
        "dph_profile_read": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
        "dph_profile_read_all": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
        "dph_debug_scan_buckets": (C.c_int, [vp, vp, i64, vp, i32, vp, vp]),
        "dph_debug_lmax": (C.c_int, [vp, i64, vp]),
        "dph_debug_aux": (C.c_int, [vp, i64, i64, vp, i64, vp, vp]),
        "dph_debug_mu": (C.c_int, [vp, vp]),
        "dph_index_get_aux_layout": (C.c_int, [vp, vp]),
        "dph_index_set_aux_layout": (C.c_int, [vp, vp]),
        "dph_debug_scan_time": (C.c_int, [vp, vp, i64, i32, vp]),
        "dph_debug_units": (C.c_int, [vp, vp]),
        "dph_debug_pq_coarse": (C.c_int, [vp, vp]),
        "dph_debug_pq_pass": (C.c_int, [vp, vp, vp, C.c_int]),
        "dph_debug_pq_pool": (C.c_int, [vp, vp, vp, i64, C.POINTER(i64)]),
        "dph_debug_pq_phases": (C.c_int, [vp, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]),
        "dph_profile_read_each": (C.c_int, [vp, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]),
        "dph_debug_bucket_counts": (C.c_int, [vp, i64, vp, vp]),
        "dph_debug_guided_segment": (i64, [i64, i64, C.c_int, C.c_int, vp]),
        "dph_debug_fused_tile": (i64, [i64, C.c_int, i64, vp]),
        "dph_index_create_pq": (C.c_int, [i32, i64, i32, i32, C.POINTER(vp)]),
        "dph_index_set_pq": (C.c_int, [vp, vp, vp, vp, vp, i32]),
        "dph_index_set_pq_list_sizes": (C.c_int, [vp, vp]),
        "dph_index_upload_pq_codes": (C.c_int, [vp, i64, i64, vp, vp]),
        "dph_index_get_transform": (C.c_int, [vp, vp]),
        "dph_index_create_twin": (C.c_int, [vp, C.POINTER(vp)]),
        "dph_stream_create_cu_range": (C.c_int, [i32, i32, i32, C.POINTER(vp)]),
        "dph_stream_destroy": (C.c_int, [vp]),
        "dph_search_prepare_dev": (C.c_int, [vp, vp, i64, i32, vp]),
        "dph_search_finish_dev": (C.c_int, [vp, vp, i64, i32, vp, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = the .so does not export what dph.h declares
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
EXPORTED = ["dph_abi_version", "dph_last_error", "dph_device_count", "dph_index_create", "dph_index_destroy",
            "dph_index_set_codec", "dph_index_upload_rows", "dph_index_fill_synthetic", "dph_index_set_idx2id",
            "dph_index_set_f2o", "dph_index_finalize", "dph_index_ntotal", "dph_index_dim", "dph_index_device",
            "dph_index_rows_dev", "dph_search", "dph_search_dev", "dph_search_sample_dev", "dph_union_bounds_dev",
            "dph_search_bounded_dev", "dph_search_get_stats", "dph_reconstruct",
            "dph_id2docword", "dph_rescore", "dph_rescore_dev", "dph_merge_topk_dev", "dph_merge_records_dev",
            "dph_debug_scan_buckets", "dph_debug_lmax", "dph_debug_aux", "dph_debug_mu", "dph_index_get_aux_layout", "dph_index_set_aux_layout", "dph_debug_scan_time", "dph_debug_units", "dph_debug_pq_coarse", "dph_debug_pq_pass", "dph_debug_pq_pool", "dph_debug_pq_phases", "dph_profile_read_each", "dph_debug_bucket_counts", "dph_debug_guided_segment", "dph_debug_fused_tile", "dph_index_upload_rows_async", "dph_host_alloc_pinned",
            "dph_host_free_pinned", "dph_stream_synchronize", "dph_index_fill_synthetic_kind", "dph_index_shard_stats",
            "dph_index_set_tuning", "dph_scan_counters", "dph_debug_wave_pairs", "dph_index_rehome_rows", "dph_index_gather_rows_dev", "dph_kmeans_step_dev", "dph_index_stored_rows", "dph_index_set_id_groups", "dph_ivf_assign_dev", "dph_index_assign_dev", "dph_index_make_list_major", "dph_score_vecs_dev", "dph_score_vecs_bwd_dev", "dph_dense_logits_dev",
            "dph_profile_enable", "dph_profile_read", "dph_profile_read_all", "dph_index_set_row_ids",
            "dph_index_set_ivf", "dph_search_ivf", "dph_search_ivf_dev", "dph_index_create_pq", "dph_index_set_pq",
            "dph_index_set_pq_list_sizes", "dph_index_upload_pq_codes", "dph_index_get_transform", "dph_index_create_twin",
            "dph_stream_create_cu_range", "dph_stream_destroy", "dph_search_prepare_dev", "dph_search_finish_dev"]


def _chk(rc: int):
    if rc != 0:
        raise DphError(rc, lib.dph_last_error().decode())


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Shard:
    """One range shard of the phrase dump resident on one GPU (a ``dph_index`` handle).

    Speaks the slice of the FAISS ``Index`` protocol that the reference's MIPS uses
    (/root/reference/densephrases/index.py:30-34, 200, 286): ``ntotal``, ``d``, ``search``, ``reconstruct``.
    """

    def __init__(self, n_rows: int, device: int = 0, id_base: int = 0):
        self._h = C.c_void_p()
        _chk(lib.dph_index_create(int(device), int(n_rows), int(id_base), C.byref(self._h)))
        self.device = int(device)
        self.id_base = int(id_base)
        self.n_rows = int(n_rows)            # stored rows (list-major shards: padding included)

    @classmethod
    def from_faiss_index(cls, index, device: int = 0, chunk_codes: int = 1 << 22, keep=None):
        """A PQ index resident in HBM from a parsed FAISS file (densephrases_amd.faiss_io: PreTransformIndex over an
        IVFPQIndex, or a bare IVFPQIndex): the reference's own index type, index.py:30-33.  The codes / ids of the
        inverted lists are uploaded list by list (np.memmap views of merged.invdata are read once, in chunks).  Call
        set_idx2id / set_id_groups / set_f2o as for a raw-dump shard, then finalize().
        ``keep``: callable ids -> bool mask; only those codes of every list are uploaded -- the shard of ONE rank of a
        range-sharded job (every rank holds the OPQ matrix, all centroids and codebooks, and the codes of its own id range)."""
        from .faiss_io import IVFPQIndex, PreTransformIndex
        ivf = index.index if isinstance(index, PreTransformIndex) else index
        if not isinstance(ivf, IVFPQIndex):
            raise TypeError("from_faiss_index: an IndexIVFPQ (optionally behind an IndexPreTransform) is needed")
        A = b = None
        if isinstance(index, PreTransformIndex):
            if len(index.chain) != 1 or index.chain[0].A.shape != (DIM, DIM):
                raise ValueError("from_faiss_index: one 768 x 768 linear pre-transform (the OPQ matrix) is supported")
            A = np.ascontiguousarray(index.chain[0].A, dtype=np.float32)
            b = None if index.chain[0].b is None else np.ascontiguousarray(index.chain[0].b, dtype=np.float32)
        if ivf.d != DIM or ivf.nbits != 8 or ivf.metric != 0:
            raise ValueError("from_faiss_index: d = 768, 8-bit codes and METRIC_INNER_PRODUCT are supported")
        masks = None
        if keep is not None:
            masks = [np.asarray(keep(np.asarray(i, dtype=np.int64)), dtype=bool) if len(i) else np.zeros(0, bool) for i in ivf.list_ids]
            n_local = int(sum(int(m.sum()) for m in masks))
        else:
            n_local = int(ivf.ntotal)
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        _chk(lib.dph_index_create_pq(int(device), n_local, int(ivf.nlist), int(ivf.M), C.byref(self._h)))
        self.device, self.id_base, self.n_rows = int(device), 0, n_local
        try:
            cent = np.ascontiguousarray(ivf.centroids, dtype=np.float32)
            pqc = np.ascontiguousarray(ivf.pq_centroids, dtype=np.float32)
            _chk(lib.dph_index_set_pq(self._h, _p(A), _p(b), _p(cent), _p(pqc), 1 if ivf.by_residual else 0))
            sizes = np.asarray([len(i) for i in ivf.list_ids] if masks is None else [int(m.sum()) for m in masks], dtype=np.int64)
            _chk(lib.dph_index_set_pq_list_sizes(self._h, _p(sizes)))
            pos = 0
            for li, (codes, ids) in enumerate(zip(ivf.list_codes, ivf.list_ids)):
                if masks is not None:
                    if not masks[li].any():
                        continue
                    codes, ids = np.asarray(codes)[masks[li]], np.asarray(ids)[masks[li]]
                n = len(ids)
                for o in range(0, n, chunk_codes):
                    c = np.ascontiguousarray(codes[o:o + chunk_codes], dtype=np.uint8)
                    i = np.ascontiguousarray(ids[o:o + chunk_codes], dtype=np.int64)
                    _chk(lib.dph_index_upload_pq_codes(self._h, pos + o, len(i), _p(c), _p(i)))
                pos += n
        except Exception:
            self.close()                    # a half-loaded index holds gigabytes of HBM
            raise
        self.pq = {"nlist": int(ivf.nlist), "M": int(ivf.M), "nprobe": 256}
        return self

    def transform(self) -> np.ndarray:
        """A [768,768] of the index's pre-transform x' = A x: what index.py:32 reads as ``self.R`` (identity on raw-dump shards)"""
        out = np.empty((DIM, DIM), dtype=np.float32)
        _chk(lib.dph_index_get_transform(self._h, _p(out)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            for t in list(getattr(self, "_twins", [])):      # twins share this index's rows: they go first
                t.close()
            self._twins = []
            rc = lib.dph_index_destroy(self._h)
            if rc != 0:                                      # refused (twins alive elsewhere): the handle -- and its HBM -- stay ours to retry
                raise DphError(rc, lib.dph_last_error().decode())
            self._h = C.c_void_p()
            parent = getattr(self, "_twin_of", None)
            if parent is not None and self in getattr(parent, "_twins", []):
                parent._twins.remove(self)
            self._twin_of = None

    def twin(self) -> "Shard":
        """A second handle over the same rows, metadata and shard constants with search scratch of its own: one handle per batch
        in flight (dph_index_create_twin; dist.PipelinedSearcher).  Closed with -- or before -- this shard."""
        t = Shard.__new__(Shard)
        t._h = C.c_void_p()
        _chk(lib.dph_index_create_twin(self._h, C.byref(t._h)))
        for name in ("device", "id_base", "n_rows"):
            setattr(t, name, getattr(self, name))
        t._twin_of = self
        if not hasattr(self, "_twins"):
            self._twins = []
        self._twins.append(t)
        return t

    def search_prepare_dev(self, x_ptr: int, n: int, k: int, stream: int = 0):
        """First stage of a search in two stages: quantise + the sampled levels (dph_search_prepare_dev)."""
        _chk(lib.dph_search_prepare_dev(self._h, C.c_void_p(x_ptr), int(n), int(k), C.c_void_p(stream)))

    def search_finish_dev(self, x_ptr: int, n: int, k: int, D_ptr: int, I_ptr: int, status_ptr: int, stream: int = 0):
        """Second stage: full scan, refine, select, retry chain -- the result of dph_search_dev (dph_search_finish_dev)."""
        _chk(lib.dph_search_finish_dev(self._h, C.c_void_p(x_ptr), int(n), int(k), C.c_void_p(D_ptr), C.c_void_p(I_ptr),
                                       C.c_void_p(status_ptr), C.c_void_p(stream)))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- faiss-like attributes
    @property
    def ntotal(self) -> int:
        return int(lib.dph_index_ntotal(self._h))

    @property
    def d(self) -> int:
        return int(lib.dph_index_dim(self._h))

    @property
    def handle(self):
        return self._h

    # ---- loading
    def set_codec(self, offset: float, scale: float):
        _chk(lib.dph_index_set_codec(self._h, float(offset), float(scale)))

    def upload(self, rows: np.ndarray, row0: int = 0):
        rows = np.ascontiguousarray(rows, dtype=np.int8)
        assert rows.ndim == 2 and rows.shape[1] == DIM
        _chk(lib.dph_index_upload_rows(self._h, int(row0), int(rows.shape[0]), _p(rows)))

    def fill_synthetic(self, seed: int = 42, stream: int = 0, kind: int = 0):
        """kind 0 = the i.i.d. dump of BASELINE config 2, 1 = mixture of 4096 Gaussians + saturated outlier rows, 2 =
        document-ordered runs of near-duplicates, 3 = the mixture without outliers (csrc/dph_quant.hip)."""
        _chk(lib.dph_index_fill_synthetic_kind(self._h, int(seed), int(kind), C.c_void_p(stream)))

    def upload_async(self, pinned_ptr: int, row0: int, n: int, stream: int = 0):
        """rows [row0, row0+n) from PINNED host memory (dph_host_alloc_pinned), asynchronous on `stream`."""
        _chk(lib.dph_index_upload_rows_async(self._h, int(row0), int(n), C.c_void_p(pinned_ptr), C.c_void_p(stream)))

    def set_tuning(self, key: str, *values: int):
        arr = np.asarray(values, dtype=np.int32)
        _chk(lib.dph_index_set_tuning(self._h, key.encode(), _p(arr) if arr.size else None, int(arr.size)))

    def shard_stats(self) -> dict:
        r, ra, no = C.c_double(0), C.c_double(0), C.c_int(0)
        _chk(lib.dph_index_shard_stats(self._h, C.byref(r), C.byref(ra), C.byref(no)))
        return {"rmax": r.value, "rmax_all": ra.value, "n_outliers": no.value}

    AUX_LAYOUT_INTS = 28

    def aux_layout(self) -> np.ndarray:
        """int32[28]: stride (0 none / 4 norm codes / 32 norm codes + replicas), norm slots, replica slots, low-digit clamp, the
        dimension of every replica slot (include/dph.h dph_index_get_aux_layout)."""
        out = np.zeros(self.AUX_LAYOUT_INTS, dtype=np.int32)
        _chk(lib.dph_index_get_aux_layout(self._h, _p(out)))
        return out

    def set_aux_layout(self, layout):
        """Impose a layout (a sharded job: the one all ranks agreed on); the aux rows are rebuilt to it."""
        arr = np.ascontiguousarray(layout, dtype=np.int32)
        assert arr.size == self.AUX_LAYOUT_INTS
        _chk(lib.dph_index_set_aux_layout(self._h, _p(arr)))

    def scan_counters(self):
        """(pairs emitted, wave-level emit-path triggers) of the last scan launch (synchronises the device)."""
        a, b = C.c_int64(0), C.c_int64(0)
        _chk(lib.dph_scan_counters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def gather_rows_dev(self, idx_ptr: int, m: int, out_ptr: int, stream: int = 0):
        """int8 rows idx[0..m) of the resident shard -> a contiguous device buffer [m,768] (k-means training sample)"""
        _chk(lib.dph_index_gather_rows_dev(self._h, C.c_void_p(idx_ptr), int(m), C.c_void_p(out_ptr), C.c_void_p(stream)))

    def kmeans_step_dev(self, rows_ptr: int, m: int, centroids_ptr: int, nlist: int, assign_ptr: int, gap_ptr: int, counts_ptr: int,
                        bias_ptr: int = 0, spherical: bool = True, stream: int = 0):
        """one Lloyd iteration over int8 rows [m,768] (dph_kmeans_step_dev): MFMA assignment + integer-sum centroid update"""
        _chk(lib.dph_kmeans_step_dev(self._h, C.c_void_p(rows_ptr), int(m), C.c_void_p(centroids_ptr), int(nlist),
                                     C.c_void_p(bias_ptr) if bias_ptr else None, int(bool(spherical)), C.c_void_p(assign_ptr),
                                     C.c_void_p(gap_ptr), C.c_void_p(counts_ptr), C.c_void_p(stream)))

    def rehome_rows(self, stream: int = 0):
        """move the resident rows into a freshly allocated buffer (dph_index_rehome_rows)"""
        _chk(lib.dph_index_rehome_rows(self._h, C.c_void_p(stream)))

    def wave_pairs(self, image: int = 0) -> np.ndarray:
        """pairs every scan wave emitted in the last scan launch of the first attempt (image 0) / the retry (image 1)."""
        out = np.zeros(4096, dtype=np.uint32)
        n = C.c_int(0)
        _chk(lib.dph_debug_wave_pairs(self._h, int(image), _p(out), out.size, C.byref(n)))
        return out[:n.value]

    def set_idx2id(self, doc: np.ndarray, word: np.ndarray):
        doc = np.ascontiguousarray(doc, dtype=np.int32)
        word = np.ascontiguousarray(word, dtype=np.int32)
        assert doc.shape == word.shape == (self.ntotal,)
        _chk(lib.dph_index_set_idx2id(self._h, _p(doc), _p(word)))

    def set_f2o(self, doc_ids: np.ndarray, f2o_off: np.ndarray, f2o: np.ndarray):
        doc_ids = np.ascontiguousarray(doc_ids, dtype=np.int32)
        f2o_off = np.ascontiguousarray(f2o_off, dtype=np.int64)
        f2o = np.ascontiguousarray(f2o, dtype=np.int32)
        assert f2o_off.shape == (doc_ids.shape[0] + 1,)
        _chk(lib.dph_index_set_f2o(self._h, int(doc_ids.shape[0]), _p(doc_ids), _p(f2o_off), _p(f2o)))

    def set_id_groups(self, id_offsets, row_starts):
        """Sub-index groups of a merged index: ids = id_offsets[g] + (row - row_starts[g]) (dph.h)."""
        off = np.ascontiguousarray(id_offsets, dtype=np.int64)
        rs = np.ascontiguousarray(row_starts, dtype=np.int64)
        assert rs.shape[0] == off.shape[0] + 1 or off.shape[0] == 0
        _chk(lib.dph_index_set_id_groups(self._h, int(off.shape[0]), _p(off) if off.size else None, _p(rs) if off.size else None))

    def set_row_ids(self, row_ids: np.ndarray, n_ids: int):
        """List-major shard: global id of every stored row (-1 = list padding)."""
        row_ids = np.ascontiguousarray(row_ids, dtype=np.int64)
        _chk(lib.dph_index_set_row_ids(self._h, _p(row_ids), int(n_ids)))

    def set_ivf(self, centroids: np.ndarray, tile_list: np.ndarray):
        centroids = np.ascontiguousarray(centroids, dtype=np.float32)
        tile_list = np.ascontiguousarray(tile_list, dtype=np.int32)
        assert centroids.ndim == 2 and centroids.shape[1] == DIM
        _chk(lib.dph_index_set_ivf(self._h, int(centroids.shape[0]), _p(centroids), _p(tile_list)))

    def search_ivf(self, x: np.ndarray, k: int, nprobe: int):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        _chk(lib.dph_search_ivf(self._h, _p(x), n, int(k), int(nprobe), _p(D), _p(I)))
        return D, I

    def finalize(self, stream: int = 0):
        _chk(lib.dph_index_finalize(self._h, C.c_void_p(stream)))
        self._aux_synced = 0           # (the shard may have chosen a new aux layout: the ranks of a sharded job agree again, dist.sync_aux_layout)

    def rows_dev_ptr(self) -> int:
        return int(lib.dph_index_rows_dev(self._h) or 0)

    # ---- faiss Index.search (index.py:200)
    def search(self, x: np.ndarray, k: int):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == DIM
        n = x.shape[0]
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        _chk(lib.dph_search(self._h, _p(x), n, int(k), _p(D), _p(I)))
        return D, I

    def search_dev(self, x_ptr: int, n: int, k: int, D_ptr: int, I_ptr: int, status_ptr: int, stream: int = 0):
        _chk(lib.dph_search_dev(self._h, C.c_void_p(x_ptr), int(n), int(k), C.c_void_p(D_ptr), C.c_void_p(I_ptr),
                                C.c_void_p(status_ptr), C.c_void_p(stream)))

    def search_sample_dev(self, x_ptr: int, n: int, top_ptr: int, stream: int = 0):
        """Phase 1 of the sharded search: the 16 best pre-pass sample scores per row -> int32 [n,16] (dph.h)."""
        _chk(lib.dph_search_sample_dev(self._h, C.c_void_p(x_ptr), int(n), C.c_void_p(top_ptr), C.c_void_p(stream)))

    def search_bounded_dev(self, x_ptr: int, n: int, k: int, tau_ptr: int, D_ptr: int, I_ptr: int, status_ptr: int,
                           bound_ptr: int, stream: int = 0):
        """Phase 2: scan under the caller's per-row bounds; also returns the dropped-row score bound f64 [n]."""
        _chk(lib.dph_search_bounded_dev(self._h, C.c_void_p(x_ptr), int(n), int(k), C.c_void_p(tau_ptr),
                                        C.c_void_p(D_ptr), C.c_void_p(I_ptr), C.c_void_p(status_ptr),
                                        C.c_void_p(bound_ptr), C.c_void_p(stream)))

    def search_ivf_dev(self, x_ptr: int, n: int, k: int, nprobe: int, D_ptr: int, I_ptr: int, status_ptr: int,
                       stream: int = 0):
        _chk(lib.dph_search_ivf_dev(self._h, C.c_void_p(x_ptr), int(n), int(k), int(nprobe), C.c_void_p(D_ptr),
                                    C.c_void_p(I_ptr), C.c_void_p(status_ptr), C.c_void_p(stream)))

    def stats(self) -> dict:
        s = SearchStats()
        _chk(lib.dph_search_get_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in SearchStats._fields_}

    def profile_enable(self, on: bool = True):
        _chk(lib.dph_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        """(summed scan-kernel milliseconds, launches) since the last read; synchronises the recorded events."""
        ms, cnt = C.c_double(0.0), C.c_int(0)
        _chk(lib.dph_profile_read(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def profile_read_each(self, which: int = 0, cap: int = 4096):
        """per-launch milliseconds of the bracketed launches since the last read (0: full scans / PQ coarse filter scans, 1: ladder levels)"""
        out, n = np.zeros(cap, dtype=np.float64), C.c_int(0)
        _chk(lib.dph_profile_read_each(self._h, int(which), _p(out), int(cap), C.byref(n)))
        return out[:min(n.value, cap)].copy()

    def profile_read_all(self):
        """(full-scan ms, full-scan launches, ladder-scan ms, ladder-scan launches) since the last read."""
        ms, cnt, lms, lcnt = C.c_double(0.0), C.c_int(0), C.c_double(0.0), C.c_int(0)
        _chk(lib.dph_profile_read_all(self._h, C.byref(ms), C.byref(cnt), C.byref(lms), C.byref(lcnt)))
        return ms.value, cnt.value, lms.value, lcnt.value

    def debug_scan_buckets(self, x: np.ndarray, tau: Optional[np.ndarray] = None, tile_stride: int = 1):
        """One filter-scan launch + refine for n <= 256 query rows: list of (scores int32, rows uint32) per query row
        and a bool array `lost` (dph.h: dph_debug_scan_buckets)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        cap = 32768
        keys = np.zeros((n, cap), dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint32)
        t = None if tau is None else np.ascontiguousarray(tau, dtype=np.int32)
        _chk(lib.dph_debug_scan_buckets(self._h, _p(x), n, _p(t), int(tile_stride), _p(keys), _p(counts)))
        out = []
        for q in range(n):
            kq = keys[q, :int(counts[q] & 0x7FFFFFFF)]
            score = ((kq >> np.uint64(32)).astype(np.uint32) ^ np.uint32(0x80000000)).view(np.int32)
            rows = np.uint32(0xFFFFFFFF) - (kq & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            out.append((score, rows))
        return out, (counts >> np.uint32(31)).astype(bool)

    def debug_lmax(self, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=np.int32)
        _chk(lib.dph_debug_lmax(self._h, int(n), _p(out)))
        return out

    def debug_aux(self, row0: int = 0, n_rows: int = 0, n_q: int = 0):
        """(aux rows [n_rows, stride] int8 | None, aux digits of the last quantised query rows [n_q, 32] int8 | None,
        {"stride", "unit", "q2max", "n_rep"})"""
        info = np.zeros(4, dtype=np.int32)
        _chk(lib.dph_debug_aux(self._h, 0, 0, None, 0, None, _p(info)))
        stride = int(info[0])
        aux = np.zeros((n_rows, stride), dtype=np.int8) if (n_rows and stride) else None
        qaux = np.zeros((n_q, 32), dtype=np.int8) if n_q else None
        _chk(lib.dph_debug_aux(self._h, int(row0), int(n_rows) if aux is not None else 0, _p(aux) if aux is not None else None,
                               int(n_q), _p(qaux) if qaux is not None else None, _p(info)))
        return aux, qaux, {"stride": stride, "unit": int(info[1]), "q2max": int(info[2]), "n_rep": int(info[3])}

    def debug_mu(self) -> np.ndarray:
        out = np.zeros(768, dtype=np.int32)
        _chk(lib.dph_debug_mu(self._h, _p(out)))
        return out

    def debug_bucket_counts(self, n: int):
        """(keys per query row in the buckets of the last pass, pair-pool overflow flags)"""
        raw, ov = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
        _chk(lib.dph_debug_bucket_counts(self._h, int(n), _p(raw), _p(ov)))
        return raw, ov

    def debug_pq_coarse(self):
        """(failed over to the bf16x3 chain?, candidates the filter GEMM emitted) of the last pass of a PQ index's coarse quantizer"""
        out = np.zeros(2, dtype=np.uint32)
        _chk(lib.dph_debug_pq_coarse(self._h, _p(out)))
        return (None if out[0] == 0xFFFFFFFF else bool(out[0])), int(out[1])

    def debug_pq_pass(self, n_rows: int):
        """(info dict, per_row [n_rows, 3] = candidates appended / overflow flag / bound key) of the last pass of a PQ index's ADC scan"""
        info = np.zeros(8, dtype=np.int32)
        per_row = np.zeros((max(int(n_rows), 1), 3), dtype=np.uint32)
        _chk(lib.dph_debug_pq_pass(self._h, _p(info), _p(per_row), int(n_rows)))
        keys = ("pairs", "items_taken", "units", "_", "cand_cap", "unit_cap", "pair_cap", "scratch_rows")
        return {k: int(v) for k, v in zip(keys, info) if k != "_"}, per_row[: int(n_rows)]

    def debug_pq_phases(self, which: int = 0, cap_wgs: int = 1024):
        """[records, 8] uint64 of a phase clock of the PQ chain (dph_debug_pq_phases: 0 = ADC scan workgroups, 1 = probe selection
        rows); the first call arms it"""
        out = np.zeros((cap_wgs, 8), dtype=np.uint64)
        n = C.c_int(0)
        _chk(lib.dph_debug_pq_phases(self._h, int(which), _p(out), int(cap_wgs), C.byref(n)))
        return out[: min(int(n.value), cap_wgs)]

    def debug_pq_pool(self, cap: int = 1 << 20):
        """(lists, score keys as fp32, query rows) of the candidate pool the coarse filter of the last pass left behind"""
        lk = np.zeros((cap, 2), dtype=np.uint32)
        q = np.zeros(cap, dtype=np.uint16)
        n = C.c_int64(0)
        _chk(lib.dph_debug_pq_pool(self._h, _p(lk), _p(q), int(cap), C.byref(n)))
        m = min(int(n.value), cap)
        key = lk[:m, 1]
        bits = np.where(key & np.uint32(0x80000000), key & np.uint32(0x7FFFFFFF), ~key)
        return lk[:m, 0].astype(np.int64), bits.view(np.float32), q[:m].astype(np.int64)

    def debug_scan_time(self, x: np.ndarray, iters: int = 5) -> np.ndarray:
        """Durations (ms) of `iters` full filter-scan launches for the first <= 256 rows of x under a bound nothing reaches."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(iters, dtype=np.float32)
        _chk(lib.dph_debug_scan_time(self._h, _p(x), int(x.shape[0]), int(iters), _p(out)))
        return out

    def assign_lists_dev(self, centroids_ptr: int, nlist: int, best_ptr: int, gap_ptr: int, row0: int = 0,
                         n: Optional[int] = None, bias_ptr: int = 0, stream: int = 0):
        """arg-max_l <x_r, c_l> (+ bias_l) for rows [row0, row0+n) of the resident shard (dph_index_assign_dev)."""
        n = self.n_rows - row0 if n is None else n
        _chk(lib.dph_index_assign_dev(self._h, int(row0), int(n), C.c_void_p(centroids_ptr), int(nlist),
                                      C.c_void_p(bias_ptr) if bias_ptr else None, C.c_void_p(best_ptr), C.c_void_p(gap_ptr),
                                      C.c_void_p(stream)))

    def make_list_major(self, assign_ptr: int, centroids: np.ndarray, stream: int = 0):
        """flat resident shard -> list-major IVF shard on the device (dph_index_make_list_major); finalize afterwards"""
        c = np.ascontiguousarray(centroids, dtype=np.float32)
        assert c.ndim == 2 and c.shape[1] == DIM
        _chk(lib.dph_index_make_list_major(self._h, C.c_void_p(assign_ptr), int(c.shape[0]), _p(c), C.c_void_p(stream)))
        self.n_rows = int(lib.dph_index_stored_rows(self._h))      # stored rows changed (list padding)

    def debug_units(self) -> dict:
        out = np.zeros(4, dtype=np.int32)
        _chk(lib.dph_debug_units(self._h, _p(out)))
        return {"chunks": int(out[0]), "units": int(out[1]), "cap_error": int(out[2]), "taken_by_full_scan": int(out[3])}

    # ---- faiss reconstruct (index.py:31,286)
    def reconstruct(self, idx: int) -> np.ndarray:
        out = np.empty(DIM, dtype=np.float32)
        _chk(lib.dph_reconstruct(self._h, int(idx), _p(out)))
        return out

    # ---- MIPS.get_idxs (index.py:124-141)
    def id2docword(self, I: np.ndarray):
        I = np.ascontiguousarray(I, dtype=np.int64)
        doc = np.empty(I.shape, dtype=np.int32)
        word = np.empty(I.shape, dtype=np.int32)
        _chk(lib.dph_id2docword(self._h, _p(I), int(I.size), _p(doc), _p(word)))
        return doc, word

    # ---- window re-score (index.py:323-370)
    def rescore(self, direction: int, qhalf: np.ndarray, k: int, L: int, ids: np.ndarray, doc: np.ndarray,
                word: np.ndarray, first: np.ndarray, want_vecs: bool = False):
        qhalf = np.ascontiguousarray(qhalf, dtype=np.float32)
        n_q = qhalf.shape[0]
        nc = n_q * k
        ids = np.ascontiguousarray(ids, dtype=np.int64).reshape(nc)
        doc = np.ascontiguousarray(doc, dtype=np.int32).reshape(nc)
        word = np.ascontiguousarray(word, dtype=np.int32).reshape(nc)
        first = np.ascontiguousarray(first, dtype=np.float32).reshape(nc)
        pred = np.empty(nc, dtype=np.int32)
        best = np.empty(nc, dtype=np.float64)
        arg = np.empty(nc, dtype=np.int32)
        vecs = np.empty((nc, 2, DIM), dtype=np.float32) if want_vecs else None
        _chk(lib.dph_rescore(self._h, int(direction), _p(qhalf), n_q, int(k), int(L), _p(ids), _p(doc), _p(word),
                             _p(first), _p(pred), _p(best), _p(arg), _p(vecs)))
        return pred, best, arg, vecs

    def rescore_dev(self, direction, qhalf_ptr, n_q, k, L, ids_ptr, doc_ptr, word_ptr, first_ptr, pred_ptr, best_ptr,
                    arg_ptr, vecs_ptr=0, stream=0):
        vp = C.c_void_p
        _chk(lib.dph_rescore_dev(self._h, int(direction), vp(qhalf_ptr), int(n_q), int(k), int(L), vp(ids_ptr),
                                 vp(doc_ptr) if doc_ptr else None, vp(word_ptr) if word_ptr else None, vp(first_ptr),
                                 vp(pred_ptr), vp(best_ptr), vp(arg_ptr), vp(vecs_ptr) if vecs_ptr else None, vp(stream)))


def merge_topk_dev(device, D_parts_ptr, I_parts_ptr, n_parts, n, k, D_out_ptr, I_out_ptr, src_ptr=0, stream=0,
                   part_stride_bytes=0):
    vp = C.c_void_p
    _chk(lib.dph_merge_topk_dev(int(device), vp(D_parts_ptr), vp(I_parts_ptr), int(n_parts), int(part_stride_bytes),
                                int(n), int(k),
                                vp(D_out_ptr), vp(I_out_ptr), vp(src_ptr) if src_ptr else None, vp(stream)))


def merge_records_dev(device, D_ptr, I_ptr, best_ptr, pred_ptr, status_ptr, n_parts, n, k, D_out, I_out, best_out,
                      pred_out, status_out, stream=0, part_stride_bytes=0, bound_ptr=0):
    """Merge + follow the winners into their home shard's window results + certificate, one launch (dph.h)."""
    vp = C.c_void_p
    _chk(lib.dph_merge_records_dev(int(device), vp(D_ptr), vp(I_ptr), vp(best_ptr), vp(pred_ptr), vp(status_ptr),
                                   vp(bound_ptr) if bound_ptr else None, int(n_parts), int(part_stride_bytes), int(n),
                                   int(k), vp(D_out), vp(I_out), vp(best_out), vp(pred_out), vp(status_out), vp(stream)))


def union_bounds_dev(device, top_parts_ptr, n_parts, n, tau_ptr, stream=0):
    """Per-row bound over the union of n_parts shards' samples: int32 [n_parts,n,16] -> int32 [n] (dph.h)."""
    vp = C.c_void_p
    _chk(lib.dph_union_bounds_dev(int(device), vp(top_parts_ptr), int(n_parts), int(n), vp(tau_ptr), vp(stream)))


def stream_create_cu_range(device: int, first_cu: int, n_cus: int) -> int:
    """A HIP stream (raw pointer) restricted to the CUs [first_cu, first_cu + n_cus) of the CU-mask bit order (dph_stream_create_cu_range)."""
    st = C.c_void_p()
    _chk(lib.dph_stream_create_cu_range(int(device), int(first_cu), int(n_cus), C.byref(st)))
    return int(st.value)


_CU_STREAMS = {}


def cu_range_stream(device: int, first_cu: int, n_cus: int) -> int:
    """The process's stream for that CU range (created on first use, never destroyed: safe to hand to torch)."""
    key = (int(device), int(first_cu), int(n_cus))
    if key not in _CU_STREAMS:
        _CU_STREAMS[key] = stream_create_cu_range(*key)
    return _CU_STREAMS[key]


def stream_destroy(stream: int):
    _chk(lib.dph_stream_destroy(C.c_void_p(stream)))
