"""NumPy restatement of the kernels' counter-based dropout RNG
(han_amd/csrc/han_common.h: han_rand64 = two 32x32->64 multiply-fold rounds, four
16-bit draws per call) so that the oracle can be fed exactly the masks the HIP
kernels draw."""
import numpy as np

STREAM_SEQ, STREAM_COEF, STREAM_FTS = 0, 1, 2
_M = np.uint64(0xFFFFFFFF)


def _u32(x):
    return np.asarray(x, dtype=np.uint64) & _M


def _mum(A, B):
    c = (A ^ np.uint64(0x53C5CA59)) * (B ^ np.uint64(0x74743C1B))      # < 2^64, exact in uint64
    return c & _M, c >> np.uint64(32)


def han_rand64(seed, stream, a, b):
    """Returns (x, y), two uint32 words per counter (a, b)."""
    seed = int(seed)
    lo = np.uint64(seed & 0xFFFFFFFF)
    hi = np.uint64((seed >> 32) & 0xFFFFFFFF)
    k1 = np.uint64((int(hi) + stream * 0x9E3779B9) & 0xFFFFFFFF)
    A, B = np.broadcast_arrays(_u32(a) ^ lo, _u32(b) ^ k1)
    A, B = _mum(A, B)
    A, B = _mum(A ^ hi, B ^ lo)
    x = A ^ B
    y = B ^ (A >> np.uint64(15)) ^ _u32(A << np.uint64(17))
    return x, y


def field(x, y, f):
    """16-bit field f in [0,4) of the 64-bit output (f may be an array)."""
    f = np.asarray(f)
    w = np.where((f & 2) != 0, y, x)
    return (w >> (np.uint64(16) * (f & 1).astype(np.uint64))) & np.uint64(0xFFFF)


def keep_prob32(drop):
    """The kernels compute keep = 1.f - drop in fp32."""
    return float(np.float32(1.0) - np.float32(drop))


def _thr(drop):
    return np.uint64(int(np.float32(keep_prob32(drop)) * np.float32(65536.0)))


def seq_mask(seed, n, f, K, drop, row_offset=0):
    """(K,N,F) -- layers.py:19: counter (row, f*ceil(K/4) + k//4), field k%4."""
    KQ = (K + 3) // 4
    rows = np.arange(n)[None, :, None] + row_offset
    fs = np.arange(f)[None, None, :]
    ks = np.arange(K)[:, None, None]
    x, y = han_rand64(seed, STREAM_SEQ, rows, fs * KQ + ks // 4)
    return (field(x, y, ks % 4) < _thr(drop)).astype(np.float64)


def coef_draws(seed, dst, src, K, drop):
    """(E,K) keep draws of the attention dropout for edges dst_i <- src_j (global ids)."""
    KQ = (K + 3) // 4
    ks = np.arange(K)[None, :]
    x, y = han_rand64(seed, STREAM_COEF, np.asarray(dst)[:, None],
                      np.asarray(src)[:, None].astype(np.uint64) * KQ + ks // 4)
    return (field(x, y, ks % 4) < _thr(drop)).astype(np.float64)


def coef_mask_csr(seed, rowptr, colidx, K, drop, row_offset=0):
    """(E,K) -- layers.py:30: counter (i, j*ceil(K/4) + k//4), field k%4."""
    rows = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr)) + row_offset
    return coef_draws(seed, rows, colidx, K, drop)


def fts_mask(seed, n, d, drop, row_offset=0, slice_index=0):
    """(N,D) -- layers.py:32: counter (row, d//4), field d%4.  slice_index: column slice of a head wider than 64
    columns (HAN_FLAG_FTS_SLICE): stream 2 + 4 * slice."""
    rows = np.arange(n)[:, None] + row_offset
    ds = np.arange(d)[None, :]
    x, y = han_rand64(seed, STREAM_FTS + 4 * int(slice_index), rows, ds // 4)
    return (field(x, y, ds % 4) < _thr(drop)).astype(np.float64)


def splitmix64(x):
    m = (1 << 64) - 1
    x = (int(x) + 0x9E3779B97F4A7C15) & m
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def resolve_seed(seed, seed_dev_value=None):
    """han_resolve_seed: the effective seed of a launch that was given a device seed word."""
    if seed_dev_value is None:
        return int(seed)
    return splitmix64((int(seed) + int(seed_dev_value)) & ((1 << 64) - 1))
