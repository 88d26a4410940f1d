"""numpy restatement of the library's counter-based dropout stream (csrc/afm_common.h:
afm_make_drop / afm_keep) so parity tests can run WITH dropout against the oracle."""
import numpy as np

M32 = 0xFFFFFFFF


def _lowbias32(x):
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> 16; x = (x * 0x7feb352d) & M32
    x ^= x >> 15; x = (x * 0x846ca68b) & M32
    x ^= x >> 16
    return x


def key_of(seed: int, site: int) -> int:
    k = int(_lowbias32((seed & M32) ^ 0x9E3779B9))
    k = int(_lowbias32(k ^ ((seed >> 32) & M32)))
    k = int(_lowbias32(k ^ ((site * 0x85EBCA6B + 0x1234567) & M32)))
    return k


def keep_mask(p: float, seed: int, site: int, n: int, start: int = 0) -> np.ndarray:
    """bool[n]: keep(i) for element indices start .. start+n-1."""
    if p <= 0:
        return np.ones(n, dtype=bool)
    t = p * 4294967296.0
    thresh = M32 if t >= 4294967295.0 else int(t)
    idx = np.arange(start, start + n, dtype=np.uint64)
    lo, hi = idx & M32, idx >> 32
    h = _lowbias32(lo ^ key_of(seed, site) ^ ((hi * 0x9E3779B1) & M32))
    return h >= thresh
