"""GPU tests of what the planning computes from the RESIDENT counts (csrc/dimn_counts_dev.h): the gene statistics against pandas
to the bit, and the exact integer correlation (int8 matrix cores) against numpy's float64 evaluation of the reference's
expression (deepimpute/multinet.py:20-34: np.abs(np.corrcoef(raw.T)), NaN -> 0) and against exact rational arithmetic."""
from fractions import Fraction

import numpy as np
import pandas as pd
import pytest

from deepimpute_amd._counts import DeviceCounts

pytestmark = pytest.mark.gpu


def _counts(n, g, seed, scale=1.0, big=None):
    rng = np.random.default_rng(seed)
    u, v = rng.normal(size=(n, 6)), rng.normal(size=(g, 6))
    lam = scale * np.exp(0.8 * (u @ v.T) / np.sqrt(6) + rng.normal(0.0, 1.0, size=g))
    x = rng.poisson(lam).astype(np.float64)
    if big is not None:
        x[:, : g // 7] *= big                          # some genes in the second byte plane
    return np.minimum(x, 255.0 if big is None else 65535.0)


def _reference_abs_corr(x):
    with np.errstate(invalid="ignore", divide="ignore"):
        c = np.abs(np.corrcoef(x.T))
    return np.nan_to_num(c, nan=0.0)


@pytest.mark.parametrize("n,g,scale,big", [(700, 300, 1.0, None),          # one plane (counts < 256), ragged sizes
                                           (1000, 257, 3.0, 37.0),          # two planes
                                           (40000, 130, 1.0, 200.0),        # more than one int32 slab (32768 cells), two planes
                                           (33000, 129, 0.5, None)])        # more than one slab, one plane
def test_int8_correlation_of_resident_counts(n, g, scale, big):
    x = _counts(n, g, seed=n + g, scale=scale, big=big)
    x[:, 5] = 3.0                                      # a constant gene: NaN in numpy, 0 after fillna
    x[:, 9] = x[:, 8]                                  # a duplicated gene: correlation exactly 1
    assert (big is None) == (x.max() < 256)
    dev = DeviceCounts.try_create(x, 0)
    assert dev is not None
    pool = np.arange(g, dtype=np.int32)
    dev.corr(pool)
    got = dev.corr_read(g)
    ref = _reference_abs_corr(x)
    assert np.array_equal(got, got.T)
    assert np.abs(got - ref).max() < 5e-14             # numpy's own float64 evaluation error; the integer path is the more exact one
    assert np.all(got[5] == 0.0) and got[8, 9] == 1.0 and np.all(np.diag(got)[np.arange(g) != 5] == 1.0)
    # exact rational arithmetic on a handful of pairs: the device value is the correctly rounded quotient up to a few ulp
    xi = x.astype(np.int64)
    rng = np.random.default_rng(1)
    for i, j in rng.integers(0, g, size=(12, 2)):
        if i == 5 or j == 5:
            continue
        a, b = [int(v) for v in xi[:, i]], [int(v) for v in xi[:, j]]
        sa, sb = sum(a), sum(b)
        num = n * sum(p * q for p, q in zip(a, b)) - sa * sb
        da, db = n * sum(p * p for p in a) - sa * sa, n * sum(q * q for q in b) - sb * sb
        exact = abs(float(Fraction(num * num, da * db))) ** 0.5
        assert abs(got[i, j] - exact) <= 4 * np.spacing(max(exact, 1e-300)) + 1e-300, (i, j, got[i, j], exact)
    # a sub-pool in another order takes the same numbers
    sub = np.array([9, 3, 250 % g, 8, 5, 17], np.int32)
    dev.corr(sub)
    got_sub = dev.corr_read(sub.size)
    assert np.array_equal(got_sub, got[np.ix_(sub, sub)])
    dev.close()


def test_float64_kernel_still_takes_large_counts(monkeypatch):
    x = _counts(500, 140, seed=2)
    x[3, 7] = 70000.0                                  # beyond two byte planes
    dev = DeviceCounts.try_create(x, 0)
    dev.corr(np.arange(140, dtype=np.int32))
    got = dev.corr_read(140)
    assert np.abs(got - _reference_abs_corr(x)).max() < 5e-14
    dev.close()
    x[3, 7] = 7.0
    monkeypatch.setenv("DIMN_CORR_I8", "0")            # and the switch
    dev = DeviceCounts.try_create(x, 0)
    dev.corr(np.arange(140, dtype=np.int32))
    f64 = dev.corr_read(140)
    monkeypatch.delenv("DIMN_CORR_I8")
    dev.corr(np.arange(140, dtype=np.int32))
    i8 = dev.corr_read(140)
    assert np.abs(f64 - i8).max() < 5e-14 and np.abs(f64 - i8).max() > 0.0
    dev.close()


@pytest.mark.parametrize("n,g", [(2, 5), (7, 130), (129, 64), (8192, 70), (8200, 33), (20011, 300), (50000, 257)])
def test_device_gene_statistics_are_pandas_to_the_bit(n, g):
    x = _counts(n, g, seed=n, scale=2.0, big=(50.0 if n % 2 else None))
    frame = pd.DataFrame(x)
    dev = DeviceCounts.try_create(x, 0)
    assert dev is not None
    st = dev.gene_stats()
    assert np.array_equal(st["mean"], frame.mean().values)
    assert np.array_equal(st["var"], frame.var().values)
    assert np.array_equal(st["cmin"], x.min(axis=0)) and np.array_equal(st["cmax"], x.max(axis=0))
    assert st["vmax"] == x.max()
    dev.close()
