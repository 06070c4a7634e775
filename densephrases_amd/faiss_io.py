"""Reader and writer of the FAISS 1.6.x index files the reference's MIPS opens (index.py:30:
``faiss.read_index(index_path, faiss.IO_FLAG_ONDISK_SAME_DIR)``), without FAISS.

What the reference builds (build_phrase_index.py:96-153, 282-338) and this module reads / writes:

    IndexPreTransform "IxPT"                                   train_index :108-116
      chain[0] = OPQMatrix(768, M)  -> "LTra": A [d_out, d_in] row-major, no bias      (x' = A x)
      index    = IndexIVFPQ "IwPQ"  (quantizer IndexFlatIP "IxFI", nlist lists, M sub-quantisers x 8 bits, METRIC_INNER_PRODUCT,
                                     by_residual, DirectMap.Hashtable :138-142)
        invlists = ArrayInvertedLists "ilar"  (add_to_index output)  or
                   OnDiskInvertedLists "ilod" + <dir>/merged.invdata  (merge_indexes :282-338, what IO_FLAG_ONDISK_SAME_DIR is for)
    IndexFlatIP "IxFI" (fine_quant 'none', :117-118)

The byte layout is FAISS' ``impl/index_write.cpp`` of the 1.6 series AS REMEMBERED: FAISS itself and
every real index file are unavailable offline, so this reader is validated against this writer and against the
structure of the reference's call sites only -- **unpinned against a real file**, and said so wherever it is used.  All
integers little-endian; ``size_t``/``idx_t`` are 8 bytes; a vector is ``size_t n`` followed by n raw elements.

    index header    int32 d | int64 ntotal | int64 dummy(1<<20) | int64 dummy | uint8 is_trained | int32 metric_type (0 = IP, 1 = L2)
    "IxFI"          header | vector<float> xb
    "IxPT"          header | int32 n_transforms | transforms... | index
    "LTra"          uint8 have_bias | vector<float> A | vector<float> b | int32 d_in | int32 d_out | uint8 is_trained
    "IwPQ"          header | size_t nlist | size_t nprobe | index(quantizer) | direct_map
                    | uint8 by_residual | size_t code_size | pq | invlists
    direct_map      int8 type (0 none, 1 array, 2 hashtable) | vector<int64> array | [hashtable: vector<pair<int64,int64>>]
                    hashtable value = list_no << 32 | offset
    pq              size_t d | size_t M | size_t nbits | vector<float> centroids [M, 2^nbits, d/M]
    "ilar"          size_t nlist | size_t code_size | "full" vector<size_t> sizes | per list: codes u8 [n*code_size], ids i64 [n]
                    (or "sprs" vector<size_t> (list_no, size) pairs of the non-empty lists)
    "il00"          no inverted lists
    "ilod"          size_t nlist | size_t code_size | vector<{size_t size, capacity, offset}> | vector<{size_t offset, capacity}> free slots
                    | vector<char> filename | size_t totsize
    merged.invdata  per list at `offset`: codes u8 [capacity*code_size] then ids i64 [capacity] (the first `size` are live)
"""
from __future__ import annotations

import io
import os
import struct
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

METRIC_INNER_PRODUCT, METRIC_L2 = 0, 1
IO_FLAG_MMAP = 0x2
IO_FLAG_ONDISK_SAME_DIR = 0x8


@dataclass
class LinearTransform:
    A: np.ndarray                      # [d_out, d_in] float32: x' = A x (+ b)
    b: Optional[np.ndarray] = None     # [d_out] or None
    is_trained: bool = True

    @property
    def d_in(self):
        return int(self.A.shape[1])

    @property
    def d_out(self):
        return int(self.A.shape[0])


@dataclass
class FlatIndex:
    d: int
    xb: np.ndarray                     # [ntotal, d] float32
    metric: int = METRIC_INNER_PRODUCT

    @property
    def ntotal(self):
        return int(self.xb.shape[0])


@dataclass
class IVFPQIndex:
    d: int
    nlist: int
    M: int
    nbits: int
    centroids: np.ndarray              # [nlist, d] float32: the IndexFlatIP coarse quantizer's vectors
    pq_centroids: np.ndarray           # [M, 2^nbits, d/M] float32
    list_codes: List[np.ndarray]       # per list uint8 [n_l, M]   (np.memmap views into merged.invdata for "ilod")
    list_ids: List[np.ndarray]         # per list int64 [n_l]
    by_residual: bool = True
    metric: int = METRIC_INNER_PRODUCT
    nprobe: int = 1
    direct_map_type: int = 2           # DirectMap.Hashtable (build_phrase_index.py:141)
    ondisk_filename: Optional[str] = None

    @property
    def ntotal(self):
        return int(sum(len(i) for i in self.list_ids))

    @property
    def code_size(self):
        return self.M * ((self.nbits + 7) // 8)


@dataclass
class PreTransformIndex:
    chain: List[LinearTransform]
    index: object                      # IVFPQIndex or FlatIndex
    d: int = 0                         # the INPUT dimension (d_in of the first transform)
    is_trained: bool = True

    @property
    def ntotal(self):
        return self.index.ntotal


# ------------------------------------------------------------------------------------------------- writer
class _W:
    def __init__(self, f):
        self.f = f

    def raw(self, b):
        self.f.write(b)

    def fourcc(self, s):
        self.f.write(s.encode("ascii"))

    def pack(self, fmt, *v):
        self.f.write(struct.pack("<" + fmt, *v))

    def vec(self, a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        self.pack("Q", a.size)
        self.f.write(a.tobytes())


def _write_header(w: _W, d, ntotal, is_trained, metric):
    w.pack("i", d)
    w.pack("q", ntotal)
    w.pack("q", 1 << 20)
    w.pack("q", 1 << 20)
    w.pack("B", 1 if is_trained else 0)
    w.pack("i", metric)


def _write_flat(w: _W, ix: FlatIndex):
    w.fourcc("IxFI" if ix.metric == METRIC_INNER_PRODUCT else "IxF2")
    _write_header(w, ix.d, ix.ntotal, True, ix.metric)
    w.vec(ix.xb, np.float32)


def _write_transform(w: _W, t: LinearTransform):
    w.fourcc("LTra")
    w.pack("B", 0 if t.b is None else 1)
    w.vec(t.A, np.float32)
    w.vec(np.zeros(0, np.float32) if t.b is None else t.b, np.float32)
    w.pack("i", t.d_in)
    w.pack("i", t.d_out)
    w.pack("B", 1 if t.is_trained else 0)


def _write_ivfpq(w: _W, ix: IVFPQIndex, path: str, ondisk: bool):
    w.fourcc("IwPQ")
    _write_header(w, ix.d, ix.ntotal, True, ix.metric)
    w.pack("Q", ix.nlist)
    w.pack("Q", ix.nprobe)
    _write_flat(w, FlatIndex(ix.d, ix.centroids, ix.metric))
    # direct map
    w.pack("b", ix.direct_map_type)
    if ix.direct_map_type == 1:
        arr = np.full(ix.ntotal, -1, np.int64)
        for l, ids in enumerate(ix.list_ids):
            arr[ids] = (np.int64(l) << 32) | np.arange(len(ids), dtype=np.int64)
        w.vec(arr, np.int64)
    else:
        w.vec(np.zeros(0, np.int64), np.int64)
    if ix.direct_map_type == 2:
        pairs = np.empty((ix.ntotal, 2), np.int64)
        o = 0
        for l, ids in enumerate(ix.list_ids):
            pairs[o:o + len(ids), 0] = ids
            pairs[o:o + len(ids), 1] = (np.int64(l) << 32) | np.arange(len(ids), dtype=np.int64)
            o += len(ids)
        w.pack("Q", ix.ntotal)
        w.raw(pairs.tobytes())
    w.pack("B", 1 if ix.by_residual else 0)
    w.pack("Q", ix.code_size)
    w.pack("Q", ix.d)
    w.pack("Q", ix.M)
    w.pack("Q", ix.nbits)
    w.vec(ix.pq_centroids, np.float32)
    sizes = np.asarray([len(i) for i in ix.list_ids], np.uint64)
    if not ondisk:
        w.fourcc("ilar")
        w.pack("Q", ix.nlist)
        w.pack("Q", ix.code_size)
        # FAISS stores the list sizes densely ("full") when more than half of the lists are non-empty, else as
        # (list_no, size) pairs of the non-empty ones ("sprs")
        non0 = np.nonzero(sizes)[0]
        if len(non0) > ix.nlist // 2:
            w.fourcc("full")
            w.vec(sizes, np.uint64)
        else:
            w.fourcc("sprs")
            w.vec(np.stack([non0.astype(np.uint64), sizes[non0]], 1).reshape(-1), np.uint64)
        for codes, ids in zip(ix.list_codes, ix.list_ids):
            w.raw(np.ascontiguousarray(codes, np.uint8).tobytes())
            w.raw(np.ascontiguousarray(ids, np.int64).tobytes())
        return
    # OnDiskInvertedLists: what merge_indexes leaves next to the index (build_phrase_index.py:282-338)
    fname = ix.ondisk_filename or os.path.join(os.path.dirname(os.path.abspath(path)), "merged.invdata")
    lists = np.zeros((ix.nlist, 3), np.uint64)
    off = 0
    with open(fname, "wb") as g:
        for l, (codes, ids) in enumerate(zip(ix.list_codes, ix.list_ids)):
            n = len(ids)
            lists[l] = (n, n, off)
            g.write(np.ascontiguousarray(codes, np.uint8).tobytes())
            g.write(np.ascontiguousarray(ids, np.int64).tobytes())
            off += n * (ix.code_size + 8)
    w.fourcc("ilod")
    w.pack("Q", ix.nlist)
    w.pack("Q", ix.code_size)
    w.pack("Q", ix.nlist)
    w.raw(lists.tobytes())
    w.pack("Q", 0)                                     # no free slots
    w.vec(np.frombuffer(fname.encode("utf-8"), np.uint8), np.uint8)
    w.pack("Q", off)


def write_index(index, path: str, ondisk: bool = False):
    """``faiss.write_index`` for the index kinds above.  ``ondisk=True`` writes the inverted lists to
    ``<dir>/merged.invdata`` ("ilod"), as ``merge_indexes`` does."""
    with open(path, "wb") as f:
        w = _W(f)
        if isinstance(index, FlatIndex):
            _write_flat(w, index)
        elif isinstance(index, IVFPQIndex):
            _write_ivfpq(w, index, path, ondisk)
        elif isinstance(index, PreTransformIndex):
            w.fourcc("IxPT")
            _write_header(w, index.d or index.chain[0].d_in, index.ntotal, index.is_trained, index.index.metric)
            w.pack("i", len(index.chain))
            for t in index.chain:
                _write_transform(w, t)
            if isinstance(index.index, FlatIndex):
                _write_flat(w, index.index)
            else:
                _write_ivfpq(w, index.index, path, ondisk)
        else:
            raise TypeError(f"write_index: unsupported index {type(index).__name__}")


# ------------------------------------------------------------------------------------------------- reader
class FaissFormatError(ValueError):
    pass


class _R:
    def __init__(self, f, path):
        self.f, self.path = f, path
        self.size = os.fstat(f.fileno()).st_size

    def raw(self, n):
        # (checked against what the file still holds BEFORE reading: a damaged length field must not become an allocation)
        if n < 0 or n > self.size - self.f.tell():
            raise FaissFormatError(f"{self.path}: truncated or damaged (wanted {n} bytes at {self.f.tell()}, the file has {self.size})")
        b = self.f.read(n)
        if len(b) != n:
            raise FaissFormatError(f"{self.path}: truncated (wanted {n} bytes at {self.f.tell() - len(b)})")
        return b

    def fourcc(self):
        return self.raw(4).decode("latin-1")

    def unpack(self, fmt):
        v = struct.unpack("<" + fmt, self.raw(struct.calcsize("<" + fmt)))
        return v[0] if len(v) == 1 else v

    def vec(self, dtype):
        n = self.unpack("Q")
        if n * np.dtype(dtype).itemsize > self.size - self.f.tell():
            raise FaissFormatError(f"{self.path}: vector of {n} x {np.dtype(dtype).itemsize} B at {self.f.tell()} runs past the end of the file")
        return np.frombuffer(self.raw(n * np.dtype(dtype).itemsize), dtype=dtype).copy()


def _read_header(r: _R):
    d = r.unpack("i")
    ntotal = r.unpack("q")
    r.unpack("q")
    r.unpack("q")
    is_trained = bool(r.unpack("B"))
    metric = r.unpack("i")
    if metric > 1:
        r.unpack("f")                                  # metric_arg
    if d <= 0 or ntotal < 0:
        raise FaissFormatError(f"{r.path}: bad index header (d={d}, ntotal={ntotal})")
    return d, ntotal, is_trained, metric


def _read_index(r: _R, io_flags: int):
    h = r.fourcc()
    if h in ("IxFI", "IxF2", "IxFl"):
        d, ntotal, _, metric = _read_header(r)
        xb = r.vec(np.float32)
        if xb.size != ntotal * d:
            raise FaissFormatError(f"{r.path}: flat index holds {xb.size} floats, header says {ntotal} x {d}")
        return FlatIndex(d, xb.reshape(ntotal, d), METRIC_INNER_PRODUCT if h != "IxF2" else METRIC_L2)
    if h == "IxPT":
        d, ntotal, is_trained, _ = _read_header(r)
        nt = r.unpack("i")
        chain = []
        for _ in range(nt):
            th = r.fourcc()
            if th != "LTra":
                raise FaissFormatError(f"{r.path}: vector transform '{th}' is not a plain linear transform / OPQ matrix")
            have_bias = r.unpack("B")
            A = r.vec(np.float32)
            b = r.vec(np.float32)
            d_in, d_out = r.unpack("i"), r.unpack("i")
            trained = bool(r.unpack("B"))
            if A.size != d_in * d_out:
                raise FaissFormatError(f"{r.path}: transform matrix has {A.size} entries for {d_out} x {d_in}")
            chain.append(LinearTransform(A.reshape(d_out, d_in), b if have_bias else None, trained))
        sub = _read_index(r, io_flags)
        return PreTransformIndex(chain, sub, d, is_trained)
    if h == "IwPQ":
        d, ntotal, _, metric = _read_header(r)
        nlist, nprobe = r.unpack("Q"), r.unpack("Q")
        quant = _read_index(r, io_flags)
        if not isinstance(quant, FlatIndex) or quant.ntotal != nlist:
            raise FaissFormatError(f"{r.path}: coarse quantizer must be a flat index of nlist = {nlist} vectors")
        dm_type = r.unpack("b")
        r.vec(np.int64)                                # array direct map (rebuilt from the lists here)
        if dm_type == 2:
            n_pairs = r.unpack("Q")
            if 16 * n_pairs > r.size - r.f.tell():
                raise FaissFormatError(f"{r.path}: direct map of {n_pairs} pairs runs past the end of the file")
            r.f.seek(16 * n_pairs, io.SEEK_CUR)        # hashtable pairs: id -> list_no << 32 | offset, rebuilt from the lists
        elif dm_type not in (0, 1):
            raise FaissFormatError(f"{r.path}: unknown direct map type {dm_type}")
        by_residual = bool(r.unpack("B"))
        code_size = r.unpack("Q")
        pq_d, M, nbits = r.unpack("Q"), r.unpack("Q"), r.unpack("Q")
        pqc = r.vec(np.float32)
        if nbits != 8 or pq_d != d or d % M or pqc.size != (1 << nbits) * d or code_size != M:
            raise FaissFormatError(f"{r.path}: unsupported product quantizer (d={pq_d}, M={M}, nbits={nbits}, code_size={code_size})")
        codes, ids, fname = _read_invlists(r, nlist, code_size, io_flags)
        ix = IVFPQIndex(d, nlist, M, nbits, quant.xb, pqc.reshape(M, 1 << nbits, d // M), codes, ids, by_residual, metric,
                        nprobe, dm_type, fname)
        if ix.ntotal != ntotal:
            raise FaissFormatError(f"{r.path}: inverted lists hold {ix.ntotal} codes, header says {ntotal}")
        return ix
    raise FaissFormatError(f"{r.path}: index type '{h}' is not one the reference's builder writes (IxPT / IwPQ / IxFI)")


def _read_invlists(r: _R, nlist, code_size, io_flags):
    h = r.fourcc()
    if h == "il00":
        return [np.zeros((0, code_size), np.uint8)] * nlist, [np.zeros(0, np.int64)] * nlist, None
    if h == "ilar":
        nl, cs = r.unpack("Q"), r.unpack("Q")
        if nl != nlist or cs != code_size:
            raise FaissFormatError(f"{r.path}: inverted lists ({nl} x {cs} B) do not match the index ({nlist} x {code_size} B)")
        kind = r.fourcc()
        sizes = np.zeros(nlist, np.uint64)
        if kind == "full":
            sizes = r.vec(np.uint64)
        elif kind == "sprs":
            flat = r.vec(np.uint64)
            if flat.size % 2 or (flat[0::2] >= nlist).any():
                raise FaissFormatError(f"{r.path}: damaged sparse list-size table")
            sizes[flat[0::2].astype(np.int64)] = flat[1::2]
        else:
            raise FaissFormatError(f"{r.path}: unknown list-size encoding '{kind}'")
        if sizes.size != nlist or int(sizes.sum()) * (code_size + 8) > r.size - r.f.tell():
            raise FaissFormatError(f"{r.path}: the list sizes ({sizes.size} lists, {int(sizes.sum())} codes) do not fit the file")
        codes, ids = [], []
        for n in sizes.astype(np.int64):
            codes.append(np.frombuffer(r.raw(int(n) * code_size), np.uint8).reshape(int(n), code_size).copy())
            ids.append(np.frombuffer(r.raw(int(n) * 8), np.int64).copy())
        return codes, ids, None
    if h == "ilod":
        nl, cs = r.unpack("Q"), r.unpack("Q")
        if nl != nlist or cs != code_size:
            raise FaissFormatError(f"{r.path}: on-disk lists ({nl} x {cs} B) do not match the index ({nlist} x {code_size} B)")
        n = r.unpack("Q")
        if n != nlist:
            raise FaissFormatError(f"{r.path}: on-disk list table has {n} entries for {nlist} lists")
        lists = np.frombuffer(r.raw(24 * n), np.uint64).reshape(n, 3)
        n_slots = r.unpack("Q")
        if 16 * n_slots > r.size - r.f.tell():
            raise FaissFormatError(f"{r.path}: damaged free-slot table ({n_slots} slots)")
        r.raw(16 * n_slots)
        try:
            fname = bytes(r.vec(np.uint8)).decode("utf-8")
        except UnicodeDecodeError:
            raise FaissFormatError(f"{r.path}: the name of the inverted-list file is not text") from None
        tail = r.f.read(8)                             # totsize (absent in the oldest writers)
        if io_flags & IO_FLAG_ONDISK_SAME_DIR or not os.path.exists(fname):
            fname = os.path.join(os.path.dirname(os.path.abspath(r.path)), os.path.basename(fname))
        if "\0" in fname or not os.path.isfile(fname):
            raise FaissFormatError(f"{r.path}: the inverted-list file {fname!r} it names does not exist")
        # (an index nothing was added to has an empty data file, which cannot be mapped)
        data = np.memmap(fname, dtype=np.uint8, mode="r") if os.path.getsize(fname) > 0 else np.zeros(0, np.uint8)
        if len(tail) == 8 and struct.unpack("<Q", tail)[0] > data.size:
            raise FaissFormatError(f"{fname}: {data.size} bytes, the index expects {struct.unpack('<Q', tail)[0]}")
        codes, ids = [], []
        if (lists >= np.uint64(1) << np.uint64(62)).any():
            raise FaissFormatError(f"{r.path}: damaged on-disk list table")
        for size, cap, off in lists.astype(np.int64):
            size, cap, off = int(size), int(cap), int(off)
            if size > cap:
                raise FaissFormatError(f"{fname}: list at offset {off} holds {size} codes in a slot of {cap}")
            if off + cap * (code_size + 8) > data.size:
                raise FaissFormatError(f"{fname}: list at offset {off} (capacity {cap}) runs past the end of the file")
            codes.append(data[off:off + size * code_size].reshape(size, code_size))
            ids.append(data[off + cap * code_size:off + cap * code_size + size * 8].view(np.int64))
        return codes, ids, fname
    raise FaissFormatError(f"{r.path}: inverted-list type '{h}' not supported")


def read_index(path: str, io_flags: int = 0):
    """``faiss.read_index(path, io_flags)`` -> FlatIndex / IVFPQIndex / PreTransformIndex.  With
    IO_FLAG_ONDISK_SAME_DIR the on-disk lists are looked up next to ``path`` (index.py:30)."""
    with open(path, "rb") as f:
        return _read_index(_R(f, str(path)), io_flags)


def looks_like_faiss_index(path: str) -> bool:
    try:
        with open(path, "rb") as f:
            return f.read(4) in (b"IxPT", b"IwPQ", b"IxFI", b"IxF2")
    except OSError:
        return False
