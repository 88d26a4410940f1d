"""numpy restatement of the library's counter-based dropout stream (csrc/afm_common.h:
afm_make_drop / afm_keep) so parity tests can run WITH dropout against the oracle."""
import numpy as np

M32 = 0xFFFFFFFF


def _mad24(x, c):
    return ((x & 0xFFFFFF) * (c & 0xFFFFFF) + x) & M32


def _lowbias32(x):
    """afm_lowbias32 of csrc/afm_common.h: xorshift / 24-bit multiply-add mixer."""
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> 16; x = _mad24(x, 0x7b352d)
    x ^= x >> 13; x = _mad24(x, 0x6ca68b)
    x ^= x >> 16
    return x


def key_of(seed: int, site: int) -> int:
    k = int(_lowbias32((seed & M32) ^ 0x9E3779B9))
    k = int(_lowbias32(k ^ ((seed >> 32) & M32)))
    k = int(_lowbias32(k ^ ((site * 0x85EBCA6B + 0x1234567) & M32)))
    return k


def keep_mask(p: float, seed: int, site: int, n: int, start: int = 0) -> np.ndarray:
    """bool[n]: keep(i) for element indices start .. start+n-1 (afm_keep, round 4: two-level -- the full mixer once per block of 64
    consecutive elements, a four-instruction mix per element pair of the block, 16 bits per element against thresh16; kept values
    are still scaled by 1 / (1 - p))."""
    if p <= 0:
        return np.ones(n, dtype=bool)
    t16 = min(65535, int(p * 65536.0 + 0.5))
    idx = np.arange(start, start + n, dtype=np.uint64)
    blk = idx >> np.uint64(6)
    lo, hi = blk & M32, blk >> 32
    H = _lowbias32(lo ^ key_of(seed, site) ^ ((hi * 0x9E3779B1) & M32))
    pair = (idx & np.uint64(63)) >> np.uint64(1)
    h = _pair_mix((H + pair * PAIR_STRIDE) & M32)
    return np.where(idx & np.uint64(1), h >> 16, h & 0xFFFF) >= t16


PAIR_STRIDE = 0x9E3779


def _pair_mix(x):
    """afm_pair_mix of csrc/afm_common.h: the four-instruction per-pair mix of the two-level attention stream."""
    x = np.asarray(x, dtype=np.uint64) & M32
    x = _mad24(x, 0x7b352d); x ^= x >> 16
    return _mad24(x, 0x6ca68b)


def keep_mask16(p: float, seed: int, site: int, n: int, tk: int, start_row: int = 0):
    """Attention-probability stream (afm_keep16, round 4: two-level): 16 random bits per element; the full mixer once per
    score-matrix row (row = (b H + h) Tq + q), one four-instruction mix per key pair of that row.  `n` elements = n // tk rows
    of tk keys, row-major.  Returns (bool[n] keep, scale) with scale = 1/(1 - thresh16/65536)."""
    if p <= 0:
        return np.ones(n, dtype=bool), 1.0
    assert n % tk == 0
    t16 = min(65535, int(p * 65536.0 + 0.5))
    rows = np.arange(start_row, start_row + n // tk, dtype=np.uint64)
    lo, hi = rows & M32, rows >> 32
    H = _lowbias32(lo ^ key_of(seed, site) ^ ((hi * 0x9E3779B1) & M32))[:, None]
    pair = np.arange((tk + 1) // 2, dtype=np.uint64)[None, :]
    h = _pair_mix((H + ((pair & 0xFFFFFF) * PAIR_STRIDE & M32)) & M32)
    keep = np.empty((n // tk, tk), dtype=bool)
    keep[:, 0::2] = (h & 0xFFFF) >= t16
    keep[:, 1::2] = ((h >> 16) >= t16)[:, :tk // 2]
    return keep.reshape(-1), 1.0 / (1.0 - t16 / 65536.0)
