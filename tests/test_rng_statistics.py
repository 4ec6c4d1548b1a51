"""Statistical hygiene of the kernels' counter-based dropout generator (han_amd/csrc/han_common.h:
han_rand64), checked on its NumPy restatement tests/rng_ref.py (which the GPU parity tests pin bit for
bit against the kernels: tests/test_gpu_parity.py compares whole masks).  A home-made hash gets no
credit for looking random: uniformity of every 16-bit field, no correlation between neighbouring
counters (the keep bits of adjacent features / edges), between the four fields of one call (four heads)
and between the three streams at equal counters (input / attention / projected-row dropout of the same
ids), and keep rates at the thresholds the reference uses.

Thresholds: chi-square statistics are compared with mean + 6 sigma of their distribution
(df + 6 sqrt(2 df)); correlations of n samples with 6 / sqrt(n).  With a fixed seed list the tests are
deterministic; the bounds say "a sound generator fails this with probability < 1e-8 per check"."""
import numpy as np
import pytest

from tests import rng_ref

SEEDS = [0, 1, 0x0BADC0DE1234, 0x9E3779B97F4A7C15, (1 << 64) - 1]


def _draws(seed, stream, n_a=512, n_b=2048, a0=0, b0=0):
    a = (np.arange(n_a, dtype=np.uint64) + np.uint64(a0))[:, None]
    b = (np.arange(n_b, dtype=np.uint64) + np.uint64(b0))[None, :]
    x, y = rng_ref.han_rand64(seed, stream, a, b)
    return [rng_ref.field(x, y, f).astype(np.int64) for f in range(4)]      # 4 x (n_a, n_b) in [0, 65536)


def _chi2_uniform(v, bins):
    cnt = np.bincount(v.ravel() * bins // 65536, minlength=bins).astype(np.float64)
    e = v.size / bins
    return float(((cnt - e) ** 2 / e).sum())


def _corr(u, v):
    u = u.ravel().astype(np.float64) - 32767.5
    v = v.ravel().astype(np.float64) - 32767.5
    return float((u * v).mean() / np.sqrt((u * u).mean() * (v * v).mean()))


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("stream", [rng_ref.STREAM_SEQ, rng_ref.STREAM_COEF, rng_ref.STREAM_FTS])
def test_fields_are_uniform(seed, stream):
    """chi-square of each 16-bit field over 2^20 consecutive counters: 256 equal bins of the value, and the
    256 values of its low byte (a multiply-fold's low bits are its weakest)."""
    for f, v in enumerate(_draws(seed, stream)):
        bound = 255 + 6 * np.sqrt(2 * 255)
        assert _chi2_uniform(v, 256) < bound, (f, "high byte")
        assert _chi2_uniform((v & 0xFF) * 256, 256) < bound, (f, "low byte")


@pytest.mark.parametrize("seed", SEEDS)
def test_neighbouring_counters_and_fields_are_uncorrelated(seed):
    """lag-1 .. lag-3 correlation along b (adjacent features / neighbours) and along a (adjacent rows), between
    the four fields of one call (four heads of one element), and of the KEEP BITS at the reference's rate."""
    fs = _draws(seed, rng_ref.STREAM_SEQ)
    n = fs[0].size
    tol = 6.0 / np.sqrt(n)
    for f, v in enumerate(fs):
        for lag in (1, 2, 3):
            assert abs(_corr(v[:, :-lag], v[:, lag:])) < tol * 1.01, (f, "b", lag)
            assert abs(_corr(v[:-lag, :], v[lag:, :])) < tol * 1.01, (f, "a", lag)
    for i in range(4):
        for j in range(i + 1, 4):
            assert abs(_corr(fs[i], fs[j])) < tol, (i, j)
    thr = int(rng_ref._thr(0.6))
    keep = [(v < thr).astype(np.float64) for v in fs]
    p = thr / 65536.0
    for f, k in enumerate(keep):
        assert abs(k.mean() - p) < 6 * np.sqrt(p * (1 - p) / n), f
        c = ((k[:, :-1] - p) * (k[:, 1:] - p)).mean() / (p * (1 - p))
        assert abs(c) < tol * 1.01, f
    # joint distribution of the four heads' keep bits of one element: 16 cells against the product law
    code = sum((k.astype(np.int64) << i) for i, k in enumerate(keep)).ravel()
    cnt = np.bincount(code, minlength=16).astype(np.float64)
    exp = np.array([n * p ** bin(c).count("1") * (1 - p) ** (4 - bin(c).count("1")) for c in range(16)])
    assert float(((cnt - exp) ** 2 / exp).sum()) < 15 + 6 * np.sqrt(30)


@pytest.mark.parametrize("seed", SEEDS)
def test_streams_and_seeds_are_independent(seed):
    """Equal counters in the three streams (the same node ids key the input, attention and projected-row
    dropout) and under neighbouring seeds (consecutive training steps draw seed, seed + 1, ...)."""
    d = {s: _draws(seed, s, 256, 2048) for s in (rng_ref.STREAM_SEQ, rng_ref.STREAM_COEF, rng_ref.STREAM_FTS)}
    tol = 6.0 / np.sqrt(d[0][0].size)
    for s1, s2 in ((0, 1), (0, 2), (1, 2)):
        for f in range(4):
            assert abs(_corr(d[s1][f], d[s2][f])) < tol, (s1, s2, f)
    nxt = _draws((seed + 1) & ((1 << 64) - 1), rng_ref.STREAM_SEQ, 256, 2048)
    for f in range(4):
        assert abs(_corr(d[0][f], nxt[f])) < tol, f
        assert not np.array_equal(d[0][f], nxt[f])


def test_large_counters_do_not_collide():
    """The counters reach 2^24 .. 2^32 at the 10M-node config (a = row id, b = neighbour id * 2 + k/4): no
    structure in the high counter bits."""
    for a0, b0 in ((9_999_000, 19_998_000), (0xFFFF_F000, 0xFFFF_0000), (1 << 24, 1 << 25)):
        fs = _draws(7, rng_ref.STREAM_COEF, 256, 1024, a0=a0, b0=b0)
        for v in fs:
            assert _chi2_uniform(v, 64) < 63 + 6 * np.sqrt(126)
        assert abs(_corr(fs[0][:, :-1], fs[0][:, 1:])) < 6.0 / np.sqrt(fs[0].size) * 1.01
