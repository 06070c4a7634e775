"""Minimal HDF5 reader on libhdf5's C API through ctypes -- enough to load the reference's on-disk artefacts
 without h5py: ``phrase/*.hdf5`` (per-document groups, written by
/root/reference/densephrases/utils/embed_utils.py:235-246), ``idx2id.hdf5`` (build_phrase_index.py:268-276), and
``meta_compressed.pkl`` (blosc via libblosc, scripts/preprocess/compress_metadata.py:45-53).

Host-side I/O only; nothing here is on the GPU path.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import json
import os
import pickle
from collections import OrderedDict
from typing import Dict, Iterator, List, Optional

import numpy as np

from .dump import DocMeta, DocStore

_hid = C.c_int64
_LIB = None


def _find(name: str) -> Optional[str]:
    for d in (os.environ.get("DPH_HDF5_DIR"), "/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib64", "/usr/lib"):
        if d:
            for cand in (f"lib{name}.so", f"lib{name}_serial.so"):
                p = os.path.join(d, cand)
                if os.path.exists(p):
                    return p
    return ctypes.util.find_library(name)


def _lib():
    global _LIB
    if _LIB is None:
        path = _find("hdf5")
        if not path:
            raise ImportError("libhdf5.so not found (set DPH_HDF5_DIR); needed to read phrase/*.hdf5 and idx2id.hdf5")
        L = C.CDLL(path)
        L.H5open()
        proto = {
            "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]), "H5Fclose": (C.c_int, [_hid]),
            "H5Gopen2": (_hid, [_hid, C.c_char_p, _hid]), "H5Gclose": (C.c_int, [_hid]),
            "H5Gget_info": (C.c_int, [_hid, C.c_void_p]),
            "H5Lget_name_by_idx": (C.c_ssize_t, [_hid, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_char_p, C.c_size_t, _hid]),
            "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
            "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]), "H5Dclose": (C.c_int, [_hid]),
            "H5Oopen": (_hid, [_hid, C.c_char_p, _hid]), "H5Oclose": (C.c_int, [_hid]), "H5Iget_type": (C.c_int, [_hid]),
            "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]),
            "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
            "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
            "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            "H5Sselect_hyperslab": (C.c_int, [_hid, C.c_int, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
            "H5Screate_simple": (_hid, [C.c_int, C.POINTER(C.c_uint64), C.c_void_p]), "H5Sclose": (C.c_int, [_hid]),
            "H5Tget_class": (C.c_int, [_hid]), "H5Tget_size": (C.c_size_t, [_hid]), "H5Tget_sign": (C.c_int, [_hid]),
            "H5Tis_variable_str": (C.c_int, [_hid]), "H5Tcopy": (_hid, [_hid]), "H5Tset_size": (C.c_int, [_hid, C.c_size_t]),
            "H5Tset_cset": (C.c_int, [_hid, C.c_int]), "H5Tclose": (C.c_int, [_hid]),
            "H5Aopen": (_hid, [_hid, C.c_char_p, _hid]), "H5Aclose": (C.c_int, [_hid]), "H5Aget_type": (_hid, [_hid]),
            "H5Aread": (C.c_int, [_hid, _hid, C.c_void_p]), "H5Aexists": (C.c_int, [_hid, C.c_char_p]),
            "H5free_memory": (C.c_int, [C.c_void_p]),
        }
        for n, (res, args) in proto.items():
            f = getattr(L, n)
            f.restype, f.argtypes = res, args
        L._c_s1 = _hid.in_dll(L, "H5T_C_S1_g").value
        L._native_double = _hid.in_dll(L, "H5T_NATIVE_DOUBLE_g").value
        _LIB = L
    return _LIB


def _np_dtype(L, tid) -> np.dtype:
    cls, size = L.H5Tget_class(tid), L.H5Tget_size(tid)
    if cls == 0:   # H5T_INTEGER
        return np.dtype(("i" if L.H5Tget_sign(tid) == 1 else "u") + str(size))
    if cls == 1:   # H5T_FLOAT
        return np.dtype("f" + str(size))
    raise TypeError(f"unsupported HDF5 datatype class {cls}")


class H5Dataset:
    def __init__(self, did):
        self._L = _lib()
        self._id = did
        sp = self._L.H5Dget_space(did)
        nd = self._L.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_uint64 * max(nd, 1))()
        if nd > 0:
            self._L.H5Sget_simple_extent_dims(sp, dims, None)
        self._L.H5Sclose(sp)
        self.shape = tuple(int(dims[i]) for i in range(nd))
        t = self._L.H5Dget_type(did)
        self.dtype = _np_dtype(self._L, t)
        self._L.H5Tclose(t)

    def __len__(self):
        return self.shape[0] if self.shape else 0

    def read(self, start: int = 0, stop: Optional[int] = None) -> np.ndarray:
        """Rows [start, stop) along the first axis (the whole dataset by default)."""
        L = self._L
        if not self.shape:
            out = np.empty((), self.dtype)
            t = L.H5Dget_type(self._id)
            L.H5Dread(self._id, t, 0, 0, 0, out.ctypes.data_as(C.c_void_p))
            L.H5Tclose(t)
            return out
        stop = self.shape[0] if stop is None else min(stop, self.shape[0])
        n = max(stop - start, 0)
        out = np.empty((n,) + self.shape[1:], self.dtype)
        if out.size == 0:
            return out
        nd = len(self.shape)
        t = L.H5Dget_type(self._id)
        fs = L.H5Dget_space(self._id)
        st = (C.c_uint64 * nd)(start, *([0] * (nd - 1)))
        cnt = (C.c_uint64 * nd)(n, *self.shape[1:])
        L.H5Sselect_hyperslab(fs, 0, st, None, cnt, None)
        ms = L.H5Screate_simple(nd, cnt, None)
        rc = L.H5Dread(self._id, t, ms, fs, 0, out.ctypes.data_as(C.c_void_p))
        L.H5Sclose(ms), L.H5Sclose(fs), L.H5Tclose(t)
        if rc < 0:
            raise IOError("H5Dread failed")
        return out

    def read_into(self, out: np.ndarray, start: int = 0) -> None:
        """Rows [start, start + len(out)) straight into ``out`` (C-contiguous, this dataset's dtype): no intermediate
        array -- the streaming loader points it at pinned staging memory."""
        L = self._L
        n = out.shape[0]
        if n == 0:
            return
        assert out.flags["C_CONTIGUOUS"] and out.dtype == self.dtype and out.shape[1:] == self.shape[1:]
        nd = len(self.shape)
        t = L.H5Dget_type(self._id)
        fs = L.H5Dget_space(self._id)
        st = (C.c_uint64 * nd)(start, *([0] * (nd - 1)))
        cnt = (C.c_uint64 * nd)(n, *self.shape[1:])
        L.H5Sselect_hyperslab(fs, 0, st, None, cnt, None)
        ms = L.H5Screate_simple(nd, cnt, None)
        rc = L.H5Dread(self._id, t, ms, fs, 0, out.ctypes.data_as(C.c_void_p))
        L.H5Sclose(ms)
        L.H5Sclose(fs)
        L.H5Tclose(t)
        if rc < 0:
            raise IOError("H5Dread failed")

    def __getitem__(self, key):
        if isinstance(key, slice) and key.step in (None, 1):
            return self.read(key.start or 0, key.stop)
        return self.read()[key]

    def close(self):
        if self._id:
            self._L.H5Dclose(self._id)
            self._id = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class H5Group:
    def __init__(self, gid, owner=None):
        self._L = _lib()
        self._id = gid
        self._owner = owner

    def keys(self) -> List[str]:
        info = (C.c_uint64 * 4)()
        self._L.H5Gget_info(self._id, C.byref(info))
        n = int(info[1])
        out = []
        buf = C.create_string_buffer(1024)
        for i in range(n):       # H5_INDEX_NAME, H5_ITER_INC: the same string order h5py iterates in
            ln = self._L.H5Lget_name_by_idx(self._id, b".", 0, 0, i, buf, 1024, 0)
            out.append(buf.value[:ln].decode())
        return out

    def __iter__(self) -> Iterator[str]:
        return iter(self.keys())

    def __len__(self):
        info = (C.c_uint64 * 4)()
        self._L.H5Gget_info(self._id, C.byref(info))
        return int(info[1])

    def __contains__(self, name: str) -> bool:
        return self._L.H5Lexists(self._id, name.encode(), 0) > 0

    def group(self, name: str) -> "H5Group":
        gid = self._L.H5Gopen2(self._id, name.encode(), 0)
        if gid < 0:
            raise KeyError(name)
        return H5Group(gid, self)

    def dataset(self, name: str) -> H5Dataset:
        did = self._L.H5Dopen2(self._id, name.encode(), 0)
        if did < 0:
            raise KeyError(name)
        return H5Dataset(did)

    def __getitem__(self, name: str):
        """h5py-style access: the member group or dataset called ``name`` (KeyError if there is none)"""
        if name not in self:
            raise KeyError(name)
        oid = self._L.H5Oopen(self._id, name.encode(), 0)
        if oid < 0:
            raise KeyError(name)
        kind = self._L.H5Iget_type(oid)          # H5I_GROUP = 2, H5I_DATASET = 5 (H5Ipublic.h)
        self._L.H5Oclose(oid)
        if kind == 2:
            return self.group(name)
        if kind == 5:
            return self.dataset(name)
        raise KeyError(f"{name}: neither a group nor a dataset")

    def attr(self, name: str):
        L = self._L
        if L.H5Aexists(self._id, name.encode()) <= 0:
            raise KeyError(name)
        aid = L.H5Aopen(self._id, name.encode(), 0)
        t = L.H5Aget_type(aid)
        try:
            cls = L.H5Tget_class(t)
            if cls == 3:     # H5T_STRING
                if L.H5Tis_variable_str(t) > 0:
                    mt = L.H5Tcopy(L._c_s1)
                    L.H5Tset_size(mt, C.c_size_t(-1).value)
                    L.H5Tset_cset(mt, 1)
                    ptr = C.c_void_p()
                    L.H5Aread(aid, mt, C.byref(ptr))
                    val = C.string_at(ptr.value).decode("utf-8") if ptr.value else ""
                    if ptr.value:
                        L.H5free_memory(ptr)
                    L.H5Tclose(mt)
                    return val
                size = L.H5Tget_size(t)
                buf = C.create_string_buffer(size + 1)
                L.H5Aread(aid, t, buf)
                return buf.raw[:size].split(b"\x00")[0].decode("utf-8")
            out = C.c_double()
            L.H5Aread(aid, L._native_double, C.byref(out))
            return out.value
        finally:
            L.H5Tclose(t)
            L.H5Aclose(aid)

    def close(self):
        if self._id:
            self._L.H5Gclose(self._id)
            self._id = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class H5File(H5Group):
    def __init__(self, path: str):
        L = _lib()
        fid = L.H5Fopen(path.encode(), 0, 0)
        if fid < 0:
            raise IOError(f"cannot open {path}")
        self._fid = fid
        gid = L.H5Gopen2(fid, b"/", 0)
        super().__init__(gid)

    def close(self):
        super().close()
        if getattr(self, "_fid", 0):
            self._L.H5Fclose(self._fid)
            self._fid = 0


# ------------------------------------------------------------------------------------------------ blosc
_BLOSC = None


def blosc_decompress(data: bytes) -> bytes:
    """python-blosc's ``decompress`` (index.py:108-111) on libblosc."""
    global _BLOSC
    if _BLOSC is None:
        path = _find("blosc")
        if not path:
            raise ImportError("libblosc.so not found; needed for meta_compressed.pkl")
        _BLOSC = C.CDLL(path)
        _BLOSC.blosc_cbuffer_sizes.argtypes = [C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _BLOSC.blosc_decompress.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
        _BLOSC.blosc_decompress.restype = C.c_int
    nb, cb, bs = C.c_size_t(), C.c_size_t(), C.c_size_t()
    _BLOSC.blosc_cbuffer_sizes(data, C.byref(nb), C.byref(cb), C.byref(bs))
    out = C.create_string_buffer(nb.value)
    rc = _BLOSC.blosc_decompress(data, out, nb.value)
    if rc < 0:
        raise IOError("blosc_decompress failed")
    return out.raw[:rc]


# ------------------------------------------------------------------------------------------------ reference layout
class _LazyDocs:
    """doc_idx -> DocMeta, read on demand from the phrase dumps (index.py:143-156, 248-262) or, when present, from
    meta_compressed.pkl (index.py:106-122), with a small LRU."""

    def __init__(self, files: List[H5File], ranges: Optional[List[List[int]]], doc_ids: List[int], f2o: Dict[int, np.ndarray],
                 meta: Optional[dict], cache: int = 4096):
        self.files, self.ranges, self._ids, self._f2o, self.meta = files, ranges, doc_ids, f2o, meta
        self._lru: "OrderedDict[int, DocMeta]" = OrderedDict()
        self._cap = cache

    def keys(self):
        return self._ids

    def __contains__(self, d):
        return int(d) in self._f2o

    def _group(self, d: int) -> H5Group:
        name = str(d)
        if len(self.files) == 1:
            return self.files[0].group(name)
        if self.ranges is not None:
            for (a, b), f in zip(self.ranges, self.files):
                if a * 1000 <= d < b * 1000:
                    if name not in f:
                        raise ValueError("%d not found in dump list" % d)
                    return f.group(name)
        if name not in self.files[-1]:
            raise ValueError("%d not found in dump list" % d)
        return self.files[-1].group(name)

    def __getitem__(self, d):
        d = int(d)
        if d in self._lru:
            self._lru.move_to_end(d)
            return self._lru[d]
        if self.meta is not None and str(d) in self.meta:
            m = self.meta[str(d)]
            dt = m["dtypes"]
            dm = DocMeta(d, m["title"], blosc_decompress(m["context"]).decode("utf-8"),
                         np.frombuffer(blosc_decompress(m["f2o_start"]), dt["f2o_start"]),
                         np.frombuffer(blosc_decompress(m["word2char_start"]), dt["word2char_start"]),
                         np.frombuffer(blosc_decompress(m["word2char_end"]), dt["word2char_end"]))
        else:
            g = self._group(d)
            dm = DocMeta(d, g.attr("title"), g.attr("context"), self._f2o[d],
                         g.dataset("word2char_start").read(), g.dataset("word2char_end").read())
        self._lru[d] = dm
        if len(self._lru) > self._cap:
            self._lru.popitem(last=False)
        return dm


class ReferenceDump:
    """The reference's on-disk artefacts opened LAZILY (MIPS.__init__ + load_idx_f, index.py:24-88): idx2id is read
    whole (2 x 4 B x N, like the reference's "Load idx2id on memory"), the int8 rows stay on disk until
    ``iter_row_blocks`` streams a row range, f2o_start is read for the documents of a range only, document metadata
    on demand.  Speaks the part of the ``DocStore`` interface MIPS needs, plus row ranges for range-sharded loading.

    Merged indexes: ``idx2id.hdf5`` holds one group per sub-index, named by its id offset (build_phrase_index.py:
    145-153, 268-276; scripts/parallel/add_to_index.py:42-51).  The groups are concatenated in ascending offset order
    into dense stored rows; ``id_offsets`` / ``row_starts`` carry the id <-> row translation (dph_index_set_id_groups).
    """

    def __init__(self, phrase_dump_dir: str, idx2id_path: str):
        if os.path.isdir(phrase_dump_dir):                                        # index.py:90-99
            paths = sorted(os.path.join(phrase_dump_dir, n) for n in os.listdir(phrase_dump_dir) if "hdf5" in n)
        else:
            paths = [phrase_dump_dir]
        names = [os.path.splitext(os.path.basename(p))[0] for p in paths]
        ranges = None
        if names and "-" in names[0] and "dev" not in names[0]:
            ranges = [list(map(int, n.split("-"))) for n in names]
        files = [H5File(p) for p in paths]

        with_idx = H5File(idx2id_path)
        offsets = sorted((int(k) for k in with_idx.keys()))
        docs, words, starts = [], [], [0]
        for off in offsets:
            g = with_idx.group(str(off))
            docs.append(g.dataset("doc").read().astype(np.int32))
            words.append(g.dataset("word").read().astype(np.int32))
            starts.append(starts[-1] + docs[-1].shape[0])
        with_idx.close()
        self.id_offsets = np.asarray(offsets, dtype=np.int64)
        self.row_starts = np.asarray(starts, dtype=np.int64)
        self.row2doc = np.concatenate(docs) if docs else np.zeros(0, np.int32)
        self.row2word = np.concatenate(words) if words else np.zeros(0, np.int32)
        n = self.n_rows
        # runs of equal doc id are contiguous (build_phrase_index.py:233-236); a sub-index boundary also ends a run
        change = np.nonzero(np.diff(self.row2doc))[0] + 1
        cuts = np.unique(np.concatenate([[0], change, self.row_starts[1:-1]])) if n else np.zeros(0, np.int64)
        self.run_start = cuts.astype(np.int64)
        self.run_end = np.concatenate([cuts[1:], [n]]).astype(np.int64) if n else np.zeros(0, np.int64)
        self._f2o: Dict[int, np.ndarray] = {}
        self._paths, self._idx2id_path = paths, idx2id_path
        self._cache = None                    # attach_row_cache: the packed copy of one row range
        meta = None
        if "/phrase" in phrase_dump_dir:
            mp = os.path.join(phrase_dump_dir[:phrase_dump_dir.index("/phrase")], "meta_compressed.pkl")   # index.py:69-71
            if os.path.exists(mp):
                with open(mp, "rb") as f:
                    meta = pickle.load(f)
        self.docs = _LazyDocs(files, ranges, sorted(set(int(d) for d in self.row2doc[self.run_start].tolist())) if n else [],
                              self._f2o, meta)
        self.offset, self.scale = -2.0, 20.0
        if n:
            g = self.docs._group(int(self.row2doc[0]))
            try:
                self.offset, self.scale = float(g.attr("offset")), float(g.attr("scale"))
            except KeyError:
                pass

    @property
    def n_rows(self) -> int:
        return int(self.row2doc.shape[0])

    @property
    def single_dense_group(self) -> bool:
        return self.id_offsets.shape[0] <= 1 and (self.id_offsets.shape[0] == 0 or int(self.id_offsets[0]) == 0)

    def doc_starts(self) -> np.ndarray:
        """first stored row of every document run: where a range partition may cut (windows never leave a document)"""
        return self.run_start

    def _runs(self, lo: int, hi: int):
        a = int(np.searchsorted(self.run_start, lo, side="left"))
        b = int(np.searchsorted(self.run_start, hi, side="left"))
        if a < len(self.run_start) and lo < hi and self.run_start[a] != lo:
            raise ValueError("row ranges must start at a document boundary")
        return a, b

    # -------------------------------------------------------------------------------------------- packed row cache
    # The HDF5 stream of a rank's rows is bound by libhdf5's per-document first touch (group + dataset headers): minutes for
    # the ~600 k documents of one rank of eight, every process start.  A serving process that restarts reads the same bytes
    # again, so the first load may leave a PACKED copy of its row range behind -- the int8 rows as one flat file in stored
    # row order plus the f2o CSR of its documents -- and later loads of the same range stream that file (one sequential
    # read at file-system speed, no HDF5 object touched for rows or f2o).  The copy is keyed by the range and by a
    # fingerprint of the artefacts it was made from (names, sizes, mtimes of phrase/*.hdf5 and idx2id.hdf5, row count,
    # codec); anything else than an exact match is ignored and rewritten.  Off unless asked for: MIPS(cache_dir=...) or
    # DPH_DUMP_CACHE.
    def _fingerprint(self) -> dict:
        def stat(p):
            st = os.stat(p)
            return [os.path.basename(p), int(st.st_size), int(st.st_mtime_ns)]
        return {"format": 1, "n_rows": self.n_rows, "offset": self.offset, "scale": self.scale,
                "idx2id": stat(self._idx2id_path), "phrase": [stat(p) for p in self._paths],
                "id_offsets": [int(v) for v in self.id_offsets]}

    def attach_row_cache(self, cache_dir: str, lo: int, hi: int, write: bool = True, rows: bool = True) -> bool:
        """Serve ``read_rows_into`` / ``f2o_csr`` of the stored rows [lo, hi) from the packed copy under ``cache_dir`` when a
        valid one exists (returns True); otherwise, with ``write``, record what the HDF5 reads of that range deliver so
        that ``finish_row_cache`` can leave one behind.  ``rows`` = False: the f2o CSR only (a PQ index never reads the rows)."""
        self._cache = None
        if hi <= lo:
            return False
        a, b = self._runs(lo, hi)
        if b > a and int(self.run_end[b - 1]) != hi:
            raise ValueError("row ranges must end at a document boundary")
        os.makedirs(cache_dir, exist_ok=True)
        base = os.path.join(cache_dir, f"rows_{lo}_{hi}")
        # recordings that a start which died left behind (nothing else ever removes them; up to the size of the whole row range each).
        # Their names carry the HOST and the pid of the writer: only this host's are judged by /proc/<pid> -- on a shared file system
        # (or across containers with their own pid namespaces) another host's pid says nothing, and unlinking its live recording makes
        # that writer's publish step fail; a foreign recording (and one in the old, host-less format) is only reaped once it has
        # not been written to for a day
        import glob
        import socket
        import time as _time
        host = "".join(ch if ch.isalnum() or ch in "-_" else "_" for ch in socket.gethostname())[:48] or "host"
        for stale in glob.glob(glob.escape(base) + ".i8.tmp*"):
            tag = stale.rsplit(".tmp", 1)[1]
            owner, _, pid = tag.rpartition(".")
            try:
                if owner == host and pid.isdigit():
                    if int(pid) != os.getpid() and os.path.exists(f"/proc/{pid}"):
                        continue
                elif _time.time() - os.path.getmtime(stale) < 86400.0:
                    continue
                os.unlink(stale)
            except OSError:
                pass
        fp = self._fingerprint()
        c = {"lo": lo, "hi": hi, "base": base, "fp": fp, "want_rows": rows, "rows": None, "rows_valid": False, "f2o": None,
             "f2o_valid": False, "tmp": None, "covered": 0, "had": []}
        try:
            with open(base + ".json") as f:
                head = json.load(f)
            if head.get("fingerprint") == fp:
                c["had"] = list(head.get("have", []))
                if "f2o" in head.get("have", []):
                    z = np.load(base + ".f2o.npz")
                    c["f2o"], c["f2o_valid"] = (z["ids"], z["off"], z["f2o"]), True
                if rows and "rows" in head.get("have", []) and os.path.getsize(base + ".i8") == (hi - lo) * 768:
                    c["rows"], c["rows_valid"] = open(base + ".i8", "rb", buffering=0), True
        except (OSError, ValueError, KeyError):
            c["f2o"], c["f2o_valid"], c["rows"], c["rows_valid"] = None, False, None, False
        hit = c["f2o_valid"] and (c["rows_valid"] or not rows)
        if not hit and not write:
            if c["rows"] is not None:                       # (valid rows without their f2o table: nothing will read them)
                c["rows"].close()
            return False
        if rows and not c["rows_valid"]:
            c["tmp"] = base + f".i8.tmp{host}.{os.getpid()}"
            c["rows"] = open(c["tmp"], "wb")                # the loader reads its range front to back: the copy is appended
        self._cache = c
        return hit

    def abort_row_cache(self) -> None:
        """Drop a recording without publishing anything (the load failed): closes the handles, removes the temporary file."""
        c, self._cache = self._cache, None
        if c is None:
            return
        if c["rows"] is not None:
            c["rows"].close()
        if c["tmp"] is not None and os.path.exists(c["tmp"]):
            os.unlink(c["tmp"])

    def finish_row_cache(self) -> bool:
        """Publish what the reads since ``attach_row_cache`` recorded: the rows if they covered the whole range, the f2o CSR if
        it was built; a partial recording is dropped.  Returns whether everything asked for is now in place."""
        c, self._cache = self._cache, None
        if c is None:
            return False
        base, have = c["base"], []
        try:
            if c["f2o"] is not None:
                if not c["f2o_valid"]:
                    ids, off, f2o = c["f2o"]
                    np.savez(base + ".f2o.tmp.npz", ids=ids, off=off, f2o=f2o)
                    os.replace(base + ".f2o.tmp.npz", base + ".f2o.npz")
                have.append("f2o")
            if c["rows_valid"]:
                have.append("rows")
            elif c["tmp"] is not None and c["covered"] == c["hi"] - c["lo"]:
                c["rows"].close()
                c["rows"] = None
                try:
                    os.replace(c["tmp"], base + ".i8")
                    have.append("rows")
                except FileNotFoundError:             # somebody removed the recording under us: not written, the load itself is fine
                    pass
            elif not c["want_rows"] and "rows" in c["had"] and os.path.exists(base + ".i8"):
                have.append("rows")                   # an f2o-only attach leaves the rows of the same fingerprint alone
            if have:
                with open(base + ".json.tmp", "w") as f:
                    json.dump({"fingerprint": c["fp"], "have": have, "lo": c["lo"], "hi": c["hi"]}, f)
                os.replace(base + ".json.tmp", base + ".json")         # last: the header vouches for what is already in place
        finally:
            if c["rows"] is not None:
                c["rows"].close()
            c["rows"] = None
            if c["tmp"] is not None and os.path.exists(c["tmp"]):
                os.unlink(c["tmp"])
        return "f2o" in have and ("rows" in have or not c["want_rows"])

    def read_rows_into(self, out: np.ndarray, row0: int) -> None:
        """stored rows [row0, row0 + len(out)) -> out (int8 [n,768]); the range must consist of whole document runs"""
        hi = row0 + out.shape[0]
        c = self._cache
        inside = c is not None and c["lo"] <= row0 and hi <= c["hi"]
        if inside and c["rows_valid"]:
            # one sequential read straight into the caller's (pinned) buffer
            f, nbytes = c["rows"], out.shape[0] * 768
            f.seek((row0 - c["lo"]) * 768)
            dst = out if out.flags.c_contiguous else np.empty(out.shape, np.int8)
            view, got = memoryview(dst).cast("B"), 0
            while got < nbytes:
                m = f.readinto(view[got:])
                if not m:
                    raise IOError("packed row copy is shorter than its header says")
                got += m
            if dst is not out:
                out[:] = dst
            return
        self._read_rows_hdf5(out, row0)
        if inside and c["rows"] is not None and c["covered"] >= 0:
            if row0 == c["lo"] + c["covered"]:
                c["rows"].write(memoryview(np.ascontiguousarray(out)).cast("B"))
                c["covered"] += out.shape[0]
            else:
                c["covered"] = -1                     # not the front-to-back stream of a loader: nothing is published

    def _read_rows_hdf5(self, out: np.ndarray, row0: int) -> None:
        hi = row0 + out.shape[0]
        a, b = self._runs(row0, hi)
        for r in range(a, b):
            s, e = int(self.run_start[r]), int(self.run_end[r])
            if e > hi:
                raise ValueError("row ranges must end at a document boundary")
            ds = self.docs._group(int(self.row2doc[s])).dataset("start")
            words = self.row2word[s:e]
            if int(words[-1]) - int(words[0]) == e - s - 1:          # a contiguous slice of the document's rows
                ds.read_into(out[s - row0:e - row0], start=int(words[0]))
            else:                                                     # filtered index (_ftN): gather the kept rows
                out[s - row0:e - row0] = ds.read()[words]

    def iter_row_blocks(self, lo: int = 0, hi: Optional[int] = None, block: int = 1 << 18, buffers=None):
        """(row0, int8 [n,768]) blocks covering [lo, hi), each made of whole document runs and at most ``block`` rows
        (one run longer than that is yielded alone).  ``buffers``: int8 [>= block, 768] arrays to fill in turn (pinned
        staging memory of the loader); fresh arrays otherwise."""
        hi = self.n_rows if hi is None else hi
        a, b = self._runs(lo, hi)
        r, turn = a, 0
        while r < b:
            r0 = r
            s = int(self.run_start[r])
            e = int(self.run_end[r])
            while r + 1 < b and int(self.run_end[r + 1]) - s <= block:
                r += 1
                e = int(self.run_end[r])
            n = e - s
            if buffers is not None and n <= buffers[turn % len(buffers)].shape[0]:
                out = buffers[turn % len(buffers)][:n]
            else:
                out = np.empty((n, 768), np.int8)
            self.read_rows_into(out, s)
            yield s, out
            turn += 1
            r = max(r, r0) + 1

    def f2o_of(self, d: int) -> np.ndarray:
        d = int(d)
        if d not in self._f2o:
            self._f2o[d] = self.docs._group(d).dataset("f2o_start").read().astype(np.int64)
        return self._f2o[d]

    def f2o_csr(self, lo: int = 0, hi: Optional[int] = None):
        """(doc ids ascending, offsets, f2o) of the documents whose rows lie in [lo, hi): the device-side CSR"""
        hi = self.n_rows if hi is None else hi
        c = self._cache
        mine = c is not None and (c["lo"], c["hi"]) == (lo, hi)
        if mine and c["f2o_valid"]:
            return c["f2o"]
        a, b = self._runs(lo, hi)
        ids = np.array(sorted(set(int(d) for d in self.row2doc[self.run_start[a:b]].tolist())), dtype=np.int32)
        lens = np.array([len(self.f2o_of(d)) for d in ids], dtype=np.int64)
        off = np.zeros(len(ids) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        f2o = (np.concatenate([self.f2o_of(d) for d in ids]).astype(np.int32) if len(ids) else np.zeros(0, np.int32))
        if mine:
            c["f2o"] = (ids, off, f2o)
        return ids, off, f2o

    def id_groups(self, lo: int = 0, hi: Optional[int] = None):
        """(id_offsets, row_starts) of the stored rows [lo, hi) with row_starts relative to lo, or None when ids are
        simply lo + local row (one sub-index with offset 0)"""
        hi = self.n_rows if hi is None else hi
        if self.single_dense_group:
            return None
        offs, starts = [], [0]
        for g in range(len(self.id_offsets)):
            a, b = max(int(self.row_starts[g]), lo), min(int(self.row_starts[g + 1]), hi)
            if a >= b:
                continue
            offs.append(int(self.id_offsets[g]) + a - int(self.row_starts[g]))
            starts.append(starts[-1] + (b - a))
        return np.asarray(offs, np.int64), np.asarray(starts, np.int64)

    def ids_of_rows(self, rows: np.ndarray) -> np.ndarray:
        rows = np.asarray(rows, dtype=np.int64)
        g = np.searchsorted(self.row_starts, rows, side="right") - 1
        return self.id_offsets[g] + rows - self.row_starts[g]

    def rows_of_ids(self, ids: np.ndarray) -> np.ndarray:
        """stored row of every id, -1 for ids no sub-index holds"""
        ids = np.asarray(ids, dtype=np.int64)
        if self.id_offsets.shape[0] == 0:
            return np.full(ids.shape, -1, np.int64)
        g = np.clip(np.searchsorted(self.id_offsets, ids, side="right") - 1, 0, None)
        r = ids - self.id_offsets[g]
        ok = (ids >= self.id_offsets[0]) & (r < self.row_starts[g + 1] - self.row_starts[g])
        return np.where(ok, self.row_starts[g] + r, -1)

    def doc_meta(self, doc_idx: int) -> DocMeta:
        d = int(doc_idx)
        if d not in self._f2o:
            try:
                self.f2o_of(d)
            except Exception:
                raise ValueError("%d not found in dump list" % d)          # index.py:148-156
        return self.docs[d]

    def to_store(self) -> DocStore:
        """everything in host memory (tests, small dumps, the npz converter)"""
        rows = np.empty((self.n_rows, 768), np.int8)
        if self.n_rows:
            self.read_rows_into(rows, 0)
            self.f2o_csr()
        store = DocStore.__new__(DocStore)
        store.docs = self.docs
        store.offset, store.scale = self.offset, self.scale
        store.rows, store.row2doc, store.row2word = rows, self.row2doc, self.row2word
        store.id_offsets, store.row_starts = self.id_offsets, self.row_starts
        return store


def load_reference_layout(phrase_dump_dir: str, idx2id_path: str) -> DocStore:
    """The reference layout read whole into a DocStore (small dumps / tests); MIPS itself streams a ReferenceDump."""
    return ReferenceDump(phrase_dump_dir, idx2id_path).to_store()
