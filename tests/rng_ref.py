"""NumPy restatement of the kernels' counter-based dropout RNG
(han_amd/csrc/han_common.h: han_hash / han_keep) so that the oracle can be fed
exactly the masks the HIP kernels draw."""
import numpy as np

STREAM_SEQ, STREAM_COEF, STREAM_FTS = 0, 1, 2
_M = np.uint64(0xFFFFFFFF)


def _u32(x):
    return np.asarray(x, dtype=np.uint64) & _M


def han_hash(seed, stream, a, b):
    seed = int(seed)
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    a, b = _u32(a), _u32(b)
    h = _u32(a * np.uint64(0x9E3779B1)) ^ _u32(lo + np.uint64(stream) * np.uint64(0x7F4A7C15))
    h = h ^ (h >> np.uint64(16)); h = _u32(h * np.uint64(0x85EBCA6B)); h = h ^ (h >> np.uint64(13))
    h = h ^ _u32(_u32(b * np.uint64(0xC2B2AE35)) + hi)
    h = h ^ (h >> np.uint64(16)); h = _u32(h * np.uint64(0x85EBCA6B)); h = h ^ (h >> np.uint64(13))
    h = _u32(h * np.uint64(0xC2B2AE35)); h = h ^ (h >> np.uint64(16))
    return h


def keep_prob32(drop):
    """The kernels compute keep = 1.f - drop in fp32."""
    return float(np.float32(1.0) - np.float32(drop))


def keep(seed, stream, a, b, drop):
    thr = np.uint64(int(np.float32(keep_prob32(drop)) * np.float32(16777216.0)))
    return ((han_hash(seed, stream, a, b) >> np.uint64(8)) < thr)


def seq_mask(seed, n, f, K, drop, row_offset=0):
    """(K,N,F) -- layers.py:19.  One hash per (row, f, head pair): key
    (row, f*ceil(K/2) + k//2); head k uses the 16-bit field k&1."""
    KP = (K + 1) // 2
    rows = np.arange(n)[None, :, None] + row_offset
    fs = np.arange(f)[None, None, :]
    ks = np.arange(K)[:, None, None]
    h = han_hash(seed, STREAM_SEQ, rows, fs * KP + ks // 2)
    field = (h >> (np.uint64(16) * (ks % 2).astype(np.uint64))) & np.uint64(0xFFFF)
    thr16 = np.uint64(int(np.float32(keep_prob32(drop)) * np.float32(65536.0)))
    return (field < thr16).astype(np.float64)


def coef_mask_csr(seed, rowptr, colidx, K, drop, row_offset=0):
    """(E,K) -- layers.py:30, key (i, j*K + k)."""
    rows = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr)) + row_offset
    ks = np.arange(K)[None, :]
    return keep(seed, STREAM_COEF, rows[:, None], colidx[:, None].astype(np.uint64) * K + ks,
                drop).astype(np.float64)


def coef_mask_dense(seed, n, K, drop):
    """(K,N,N)."""
    i = np.arange(n)[None, :, None]
    j = np.arange(n)[None, None, :]
    ks = np.arange(K)[:, None, None]
    return keep(seed, STREAM_COEF, i, j * K + ks, drop).astype(np.float64)


def fts_mask(seed, n, d, drop, row_offset=0):
    """(N,D) -- layers.py:32, key (row, d)."""
    rows = np.arange(n)[:, None] + row_offset
    return keep(seed, STREAM_FTS, rows, np.arange(d)[None, :], drop).astype(np.float64)
