"""Stand-in modules that let the reference's own ``densephrases/index.py`` execute
UNMODIFIED (test infrastructure; used by the golden generators oracle/make_golden*.py and by
tests/test_reference_callers.py).  The reference's files are loaded from /root/reference where that exists (the build
container) and otherwise from the byte code oracle/build_ref.py compiled from them into oracle/_ref/ (the GPU box).

The reference imports h5py, faiss, blosc, spacy and ujson, none of which is installed here.
``install()`` registers minimal fakes for exactly the API surface index.py touches
(/root/reference/densephrases/index.py:5-15, 30-32, 52-66, 82-88, 100, 108-111, 200, 248-272, 286) and
then loads the reference file from where it lies.  No reference source is copied: the module object is
created from /root/reference/densephrases/index.py at run time.

* fake ``h5py.File``   : a pickle of {group: {'attrs': {...}, 'data': {name: ndarray}}}, keys iterate
                         string-sorted like HDF5 group members.
* fake ``faiss``       : ``read_index`` unpickles {'xb': int8 [N,d]} and returns an
                         IndexPreTransform-shaped object whose transform is the identity and whose
                         ``search`` is the oracle's restated IndexFlatIP (oracle.mips_oracle.flat_ip_search).
                         This is the ONE piece of the golden outputs that is a restatement rather than
                         reference code -- FAISS itself cannot be obtained offline.
* fake ``blosc``       : zlib (only ``compress``/``decompress`` round trips matter).
* fake ``spacy``       : ``English()`` + ``sentencizer`` restated in oracle/spacy_sentencizer.py (tokenizer rules that decide
                         sentence boundaries + Sentencizer.predict).
"""
from __future__ import annotations

import importlib.util
import json
import pickle
import sys
import types
import zlib

import numpy as np

REFERENCE_ROOT = "/root/reference"


def load_ref_module(modname: str):
    """The reference's module ``modname`` (a key of oracle.build_ref.REF_FILES), executed unmodified and registered in
    sys.modules: from its source under /root/reference, or from oracle/_ref/<modname>.refpyc.  FileNotFoundError if
    neither exists (tests skip)."""
    import importlib.machinery
    import os
    from oracle.build_ref import REF_FILES, bin_path
    src = os.path.join(REFERENCE_ROOT, REF_FILES[modname])
    if os.path.exists(src) and not os.environ.get("DPH_REF_FORCE_BYTECODE"):      # (the variable: rehearse the GPU box here)
        spec = importlib.util.spec_from_file_location(modname, src)
    elif os.path.exists(bin_path(modname)):
        loader = importlib.machinery.SourcelessFileLoader(modname, bin_path(modname))
        spec = importlib.util.spec_from_loader(modname, loader)
    else:
        raise FileNotFoundError(f"{modname}: neither {src} nor {bin_path(modname)} (run `python -m oracle.build_ref` "
                                "in the build container)")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def reference_available() -> bool:
    import os
    from oracle.build_ref import REF_FILES, bin_path
    return all(os.path.exists(os.path.join(REFERENCE_ROOT, rel)) or os.path.exists(bin_path(m)) for m, rel in REF_FILES.items())


# ----------------------------------------------------------------------------- h5py
class _Dataset:
    def __init__(self, arr):
        self._a = np.asarray(arr)

    def __getitem__(self, key):
        return self._a[key]

    def __len__(self):
        return len(self._a)

    @property
    def shape(self):
        return self._a.shape


class _Group:
    def __init__(self, node):
        self._n = node
        self.attrs = node.get("attrs", {})

    def __getitem__(self, key):
        v = self._n["data"][key]
        return _Group(v) if isinstance(v, dict) else _Dataset(v)

    def __contains__(self, key):
        return key in self._n["data"]

    def keys(self):
        return sorted(self._n["data"].keys())

    def __iter__(self):
        return iter(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __len__(self):
        return len(self._n["data"])


class _File(_Group):
    def __init__(self, path, mode="r"):
        with open(path, "rb") as f:
            super().__init__({"data": pickle.load(f), "attrs": {}})

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def write_fake_h5(path, groups):
    """groups: {name: {'attrs': {...}, 'data': {dataset_name: ndarray}}}"""
    with open(path, "wb") as f:
        pickle.dump(groups, f)


# ----------------------------------------------------------------------------- faiss
class _FlatSub:
    def __init__(self, xb):
        self.xb = xb
        self.nprobe = 1
        self.quantizer = None

    def reconstruct(self, i):
        i = int(i)
        if i < 0 or i >= self.xb.shape[0]:
            raise RuntimeError("fake faiss: id not found")
        return self.xb[i].astype(np.float32) / np.float32(20.0) + np.float32(-2.0)


class _Chain:
    def __init__(self, d):
        self._vt = types.SimpleNamespace(A=np.eye(d, dtype=np.float32).reshape(-1))

    def at(self, i):
        assert i == 0
        return self._vt


class _PreTransform:
    def __init__(self, xb):
        self.index = _FlatSub(xb)
        self.d = int(xb.shape[1])
        self.ntotal = int(xb.shape[0])
        self.chain = _Chain(self.d)

    def search(self, x, k):
        from oracle.mips_oracle import flat_ip_search
        D, I, _ = flat_ip_search(np.asarray(x, np.float32), self.index.xb, k)
        return D, I


class _IVFPQSub:
    """quacks like the faiss.IndexIVFPQ that ``faiss.downcast_index(self.index.index)`` / ``extract_index_ivf`` give
    index.py:31,52-62: ``reconstruct`` (raises on unknown ids), ``nprobe``, ``quantizer``"""

    def __init__(self, parsed):
        from oracle import ivfpq_oracle as P
        self._parsed, self._P = parsed, P
        self._dm = P.DirectMap(P._ivf(parsed))
        self.nprobe = 1                      # FAISS default; index.py:53,62 sets 256
        self.quantizer = types.SimpleNamespace(ntotal=P._ivf(parsed).nlist)

    def reconstruct(self, i):
        return self._P.reconstruct(self._parsed, self._dm, int(i))


class _IVFPQPreTransform:
    """faiss.IndexPreTransform over an IndexIVFPQ read from a REAL index file (densephrases_amd.faiss_io), searched by the
    oracle's restatement of FAISS' IVFPQ (oracle/ivfpq_oracle.py)"""

    def __init__(self, parsed):
        from oracle import ivfpq_oracle as P
        from densephrases_amd.faiss_io import PreTransformIndex
        self._parsed, self._P = parsed, P
        self.index = _IVFPQSub(parsed)
        self.ntotal = int(parsed.ntotal)
        if isinstance(parsed, PreTransformIndex):
            self.d = int(parsed.chain[0].d_in)
            self.chain = types.SimpleNamespace(at=lambda i: types.SimpleNamespace(A=parsed.chain[i].A.reshape(-1)))
        else:
            self.d = int(parsed.d)
            self.chain = _Chain(self.d)

    def search(self, x, k):
        return self._P.search(self._parsed, np.asarray(x, np.float32), int(k), nprobe=int(self.index.nprobe))


def _read_index(path, flags=0):
    from densephrases_amd import faiss_io
    if faiss_io.looks_like_faiss_index(path):
        parsed = faiss_io.read_index(path, flags)
        if isinstance(parsed, faiss_io.FlatIndex):
            raise NotImplementedError("refshim faiss: a bare IndexFlat file")
        return _IVFPQPreTransform(parsed)
    return _PreTransform(pickle.load(open(path, "rb"))["xb"])


def _make_faiss():
    m = types.ModuleType("faiss")
    m.IO_FLAG_ONDISK_SAME_DIR = 0x8
    m.read_index = _read_index
    m.downcast_index = lambda idx: idx
    m.downcast_VectorTransform = lambda vt: vt
    m.vector_to_array = lambda a: np.asarray(a)
    m.extract_index_ivf = lambda idx: idx.index
    m.index_cpu_to_all_gpus = lambda q: q
    return m


# ----------------------------------------------------------------------------- spacy
class _Tok:
    def __init__(self, idx):
        self.idx = idx


class _Span:
    def __init__(self, text, start):
        self.text = text
        self._start = start

    def __getitem__(self, i):
        assert i == 0
        return _Tok(self._start)


def rule_sentences(text):
    """[(sentence_text, start_char)]: oracle/spacy_sentencizer.py, the restatement of spaCy's tokenizer + sentencizer"""
    from oracle.spacy_sentencizer import sentences
    return sentences(text)


class _English:
    def create_pipe(self, name):
        return name

    def add_pipe(self, pipe):
        pass

    def __call__(self, text):
        return types.SimpleNamespace(sents=[_Span(t, s) for t, s in rule_sentences(text)])


def install(faiss_module=None, h5py_module=None, blosc_module=None):
    """Register the fakes and return the reference's index module.  ``faiss_module`` / ``h5py_module`` / ``blosc_module``
    replace the pickle-backed fakes (tests/test_reference_callers.py passes densephrases_amd.faiss_compat and the
    libhdf5 / libblosc backed stand-ins of oracle/refshim/real_io.py)."""
    h5 = types.ModuleType("h5py")
    h5.File = _File
    sys.modules["h5py"] = h5py_module if h5py_module is not None else h5
    sys.modules["faiss"] = faiss_module if faiss_module is not None else _make_faiss()
    bl = types.ModuleType("blosc")
    bl.compress = lambda b, **kw: zlib.compress(bytes(b))
    bl.decompress = lambda b: zlib.decompress(b)
    sys.modules["blosc"] = blosc_module if blosc_module is not None else bl
    sp = types.ModuleType("spacy")
    sp_lang = types.ModuleType("spacy.lang")
    sp_en = types.ModuleType("spacy.lang.en")
    sp_en.English = _English
    sys.modules["spacy"], sys.modules["spacy.lang"], sys.modules["spacy.lang.en"] = sp, sp_lang, sp_en
    uj = types.ModuleType("ujson")
    uj.__dict__.update({k: getattr(json, k) for k in ("load", "loads", "dump", "dumps")})
    sys.modules["ujson"] = uj
    # a bare package object so that `densephrases.utils.eval_utils` resolves WITHOUT running the
    # reference's densephrases/__init__.py (which would import the encoder / transformers 2.9 API)
    pkg = types.ModuleType("densephrases")
    pkg.__path__ = []
    sys.modules["densephrases"] = pkg
    utils = types.ModuleType("densephrases.utils")
    utils.__path__ = []
    sys.modules["densephrases.utils"] = utils
    load_ref_module("densephrases.utils.eval_utils")           # index.py:15 imports normalize_answer from it
    return load_ref_module("densephrases.index")
