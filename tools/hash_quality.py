"""Statistical check of the dropout stream's 32-bit mixer (csrc/afm_common.h afm_lowbias32) on sequential
counters: keep rate of the 16-bit halves, serial correlations, avalanche.  CPU only (numpy)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.dropmask import _lowbias32, keep_mask16

n = 1 << 22
TK = 1024
keep, scale = keep_mask16(0.1, 12345, 7, 2 * n, TK)       # the two-level attention stream: 8192 rows of 1024 keys
s = keep.astype(np.float64) - keep.mean()
print("keep rate", keep.mean(), "scale", scale)
for lag in (1, 2, 3, 64, TK, TK + 1, 2 * TK):              # along keys, one / two rows down, one diagonal
    print("corr lag", lag, float((s[:-lag] * s[lag:]).mean() / s.var()))
m = keep.reshape(-1, TK).astype(np.float64)
pq = m.mean()
print("row-sum / column-sum variance over binomial:", m.sum(1).var() / (TK * pq * (1 - pq)), m.sum(0).var() / (m.shape[0] * pq * (1 - pq)))
idx = np.arange(65536, dtype=np.uint64)
base = _lowbias32(idx ^ 0x9E3779B9)
worst = 0.0
for b in range(24):
    d = base ^ _lowbias32((idx ^ (1 << b)) ^ 0x9E3779B9)
    for k in range(32):
        worst = max(worst, abs(float(((d >> np.uint64(k)) & np.uint64(1)).mean()) - 0.5))
print("worst avalanche deviation from 0.5:", worst)
