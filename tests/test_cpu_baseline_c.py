"""bench.py's C CPU comparator (oracle/csrc/cpu_flat_avx512.c through oracle/cpu_baseline_c.py: test / measurement infrastructure) against
the numpy oracle: ids identical incl. exact ties in id order, padding when the database holds fewer than k rows, query counts that
are not a multiple of the kernel's 16 lanes, any thread count."""
import numpy as np
import pytest

from oracle import mips_oracle as O


def test_c_comparator_equals_the_numpy_oracle():
    from oracle import cpu_baseline_c as B
    try:
        lib = B.load()
    except Exception as e:                                       # noqa: BLE001
        pytest.skip(f"no C toolchain / OpenMP here: {e!r}")
    if not lib.cpu_flat_has_avx512():
        pytest.skip("this host has no AVX-512")
    rng = np.random.default_rng(2)
    db = (np.clip(np.rint(rng.standard_normal((20011, 768), dtype=np.float32) * 12 + 40), -128, 127) / 20.0 - 2.0).astype(np.float32)
    db[5000:5007] = db[11]                                       # exact ties
    for nq, k, threads, n in ((128, 10, 0, 20011), (5, 3, 1, 20011), (33, 64, 7, 999), (2, 10, 3, 4)):
        q = rng.normal(0, 0.5, (nq, 768)).astype(np.float32)
        q[0] = db[11]
        D, I = B.search(lib, db[:n], q, k, threads, id_base=1000)
        Dr, Ir = O.flat_ip_search_fp32_resident(q, [db[:n]], k, id_base=1000)
        kk = min(k, n)
        # fp32 sums in a different order: ids may only swap between scores closer than that
        bad = np.nonzero(I[:, :kk] != Ir[:, :kk])
        for r, c in zip(*bad):
            assert abs(float(D[r, c]) - float(Dr[r, c])) <= 2e-6 * abs(float(Dr[r, c])) + 1e-5, (nq, k, r, c)
        np.testing.assert_allclose(D[:, :kk], Dr[:, :kk], rtol=3e-6, atol=1e-5)
        if n < k:
            assert (I[:, n:] == -1).all()
        if n > 6000:
            assert list(I[0, :min(k, 8)]) == ([1011] + list(range(6000, 6007)))[:min(k, 8)]          # the tie block in id order
