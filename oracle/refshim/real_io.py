"""``h5py`` / ``blosc`` stand-ins for the reference's unmodified index.py that read REAL files -- HDF5 through libhdf5,
blosc frames through libblosc (densephrases_amd/h5.py: ctypes on the C libraries of this image) -- instead of the
pickle-backed fakes of oracle/refshim/__init__.py.  Test infrastructure (tests/test_reference_callers.py): with these and
``densephrases_amd.faiss_compat`` as ``faiss``, the reference's ``MIPS`` runs over the reference's on-disk layout with libdph
answering every FAISS call.

Only what index.py touches (/root/reference/densephrases/index.py:82-88, 100, 148-156, 246-273): ``File(path, 'r')`` as
context manager, iteration / ``in`` / ``[]`` on groups, ``dataset[:]`` / ``dataset[i]`` / ``len(dataset)``,
``group.attrs[name]``, ``close()``."""
from __future__ import annotations

import types


class _Attrs:
    def __init__(self, g):
        self._g = g

    def __getitem__(self, name):
        return self._g.attr(name)


class _Dataset:
    def __init__(self, d):
        self._d = d
        self._all = None

    def _read(self):
        if self._all is None:
            self._all = self._d.read()
        return self._all

    def __getitem__(self, key):
        return self._read()[key]

    def __len__(self):
        return len(self._d)

    @property
    def shape(self):
        return self._d.shape


class _Group:
    def __init__(self, g):
        self._g = g
        self.attrs = _Attrs(g)

    def __getitem__(self, name):
        from densephrases_amd.h5 import H5Dataset
        v = self._g[name]
        return _Dataset(v) if isinstance(v, H5Dataset) else _Group(v)

    def __contains__(self, name):
        return name in self._g

    def keys(self):
        return self._g.keys()

    def __iter__(self):
        return iter(self._g.keys())

    def __len__(self):
        return len(self._g)

    def items(self):
        return [(k, self[k]) for k in self.keys()]


class _File(_Group):
    def __init__(self, path, mode="r"):
        from densephrases_amd.h5 import H5File
        assert mode == "r", "read-only stand-in"
        super().__init__(H5File(str(path)))

    def close(self):
        self._g.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False


def h5py_module():
    m = types.ModuleType("h5py")
    m.File = _File
    return m


def blosc_module():
    from densephrases_amd.h5 import blosc_decompress
    m = types.ModuleType("blosc")
    m.decompress = blosc_decompress
    return m
