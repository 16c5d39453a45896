"""The compact table layout (lmat_common.hpp: cpt_address): (bucket, tag) must identify the canonical k-mer exactly,
since a tag match is taken as a key match (SortedDb::begin_'s compare, src/kmerdb/SortedDb.hpp:279-354, is exact)."""
import ctypes as C

import numpy as np
import pytest

from lmat_amd import capi


def _rc(x, k):
    r = 0
    for i in range(k):
        r = (r << 2) | (3 - ((x >> (2 * i)) & 3))
    return r


def _addr(lib, k, want, km):
    nb, b, t = C.c_uint64(), C.c_uint32(), C.c_uint32()
    assert lib.lmat_table_address(k, want, int(km), C.byref(nb), C.byref(b), C.byref(t)) == 0
    return nb.value, b.value, t.value


def test_address_is_a_bijection_exhaustive_k10():
    lib = capi.load_library()
    k = 10
    seen = {}
    for x in range(0, 4 ** k):
        c = min(x, _rc(x, k))
        nb, b, t = _addr(lib, k, 1, x)
        assert 1 <= t <= 65535 and b < nb
        assert seen.setdefault((b, t), c) == c      # both strands, and nothing else, share an address
    assert len(seen) == (4 ** k + 4 ** (k // 2)) // 2  # every canonical 10-mer got its own address
    assert len(set(seen.values())) == len(seen)


@pytest.mark.parametrize("k,want", [(20, 1), (20, 1 << 30), (20, 3 << 29), (20, 3 << 30), (20, 2_900_000_000), (20, 3_120_562_176), (19, 3 << 29),
                                    (18, 1), (18, 1 << 27), (16, 1), (19, 1 << 28)])
def test_address_sampled(k, want):
    """Neighbouring k-mers (one substitution apart, shifted by one base) are the near-collisions that matter."""
    lib = capi.load_library()
    rng = np.random.default_rng(k * 1000 + (want & 0xFFFF))
    seen = {}
    buckets = []
    nb = 0
    for _ in range(6000):
        x = int(rng.integers(0, 1 << 62)) & ((1 << (2 * k)) - 1)
        fam = [x, _rc(x, k), ((x << 2) | int(rng.integers(0, 4))) & ((1 << (2 * k)) - 1), x >> 2 | int(rng.integers(0, 4)) << (2 * k - 2)]
        for p in rng.integers(0, k, 4):
            fam.append(x ^ (int(rng.integers(1, 4)) << (2 * int(p))))
        for y in fam:
            c = min(y, _rc(y, k))
            nb, b, t = _addr(lib, k, want, y)
            assert 1 <= t <= 65535 and b < nb and nb >= min(want, 1 << min(32, 2 * (k - 3)))
            assert seen.setdefault((b, t), c) == c
        buckets.append(_addr(lib, k, want, x)[1])
    assert len(set(seen.values())) == len(seen)
    # the bucket hash spreads evenly: 64 equal ranges of the table, chi-square with 63 degrees of freedom
    h = np.bincount((np.array(buckets, dtype=np.float64) * 64 / nb).astype(int), minlength=64)
    chi2 = ((h - len(buckets) / 64.0) ** 2 / (len(buckets) / 64.0)).sum()
    assert chi2 < 130, chi2


def test_fractional_width_is_a_bijection_small_model():
    """The arithmetic of the fractional bucket width (cpt_geometry, wshift = -2), exhaustively on a 16-bit model of hi-space:
    bucket = hi * nb >> 16, rho = floor((hi * nb mod 2^16) / nb) -- every hi gets its own (bucket, rho), rho stays below
    ceil(2^16 / nb), and the buckets are filled evenly (floor or ceil of 2^16 / nb values each)."""
    for nb in (16385, 20000, 24576, 30001, 32769, 40000, 49152, 65535):
        hi = np.arange(1 << 16, dtype=np.uint64)
        prod = hi * np.uint64(nb)
        b = prod >> np.uint64(16)
        rho = (prod & np.uint64(0xFFFF)) // np.uint64(nb)
        assert int(b.max()) == nb - 1 and int(rho.max()) < -(-65536 // nb)
        key = b * np.uint64(8) + rho
        assert np.unique(key).size == 1 << 16
        per = np.bincount(b.astype(np.int64), minlength=nb)
        assert set(np.unique(per).tolist()) <= {65536 // nb, -(-65536 // nb)}


def test_fractional_geometry_gives_the_buckets_asked_for():
    lib = capi.load_library()
    for want in (3_120_562_176, 2_900_000_000, 3 << 29):
        nb, b, t = _addr(lib, 20, want, 0x123456789A)
        assert nb == want
    assert _addr(lib, 20, 1 << 31, 5)[0] == 1 << 31
    assert _addr(lib, 20, 1 << 30, 5)[0] == 1 << 30   # powers of two keep their integer width (the shift path of the kernels)


def test_no_compact_layout_for_short_kmers():
    lib = capi.load_library()
    nb = C.c_uint64(7)
    assert lib.lmat_table_address(8, 1 << 20, 5, C.byref(nb), None, None) == -1  # LMAT_E_ARG
    assert nb.value == 0
