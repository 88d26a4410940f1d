"""Split bf16 pair tensors (dtype AFM_BF16X2 of include/afm_hip.h): value = hi + lo, hi = bf16(v),
lo = bf16(v - hi), 16 significant bits.  The two planes of a row sit side by side, so a logical
(rows x n) tensor is a (rows x 2n) bf16 allocation; a view keeps the parent's row stride `ld` and its
lo plane ld/2 elements behind the hi plane.  The hi plane alone is an ordinary bf16 view of the same
row stride.  This class only carries pointer / shape / stride for the C ABI (ops.py); it never computes.
"""
from __future__ import annotations

import torch


class X2:
    __slots__ = ("hi", "ld")
    dtype = "bf16x2"
    is_cuda = True

    def __init__(self, hi: torch.Tensor, ld: int):
        assert hi.dtype == torch.bfloat16 and hi.dim() == 2
        assert hi.shape[0] <= 1 or hi.stride(0) == ld
        assert hi.shape[1] <= 1 or hi.stride(1) == 1
        self.hi, self.ld = hi, int(ld)

    @staticmethod
    def empty(rows: int, n: int, device) -> "X2":
        buf = torch.empty(rows, 2 * n, dtype=torch.bfloat16, device=device)
        return X2(buf[:, :n], 2 * n)

    @staticmethod
    def zeros(rows: int, n: int, device) -> "X2":
        buf = torch.zeros(rows, 2 * n, dtype=torch.bfloat16, device=device)
        return X2(buf[:, :n], 2 * n)

    @staticmethod
    def from_float(x: torch.Tensor) -> "X2":
        """Host-side split (tests / setup); the product path converts with ops.convert."""
        x = x.float()
        hi = x.to(torch.bfloat16)
        lo = (x - hi.float()).to(torch.bfloat16)
        out = X2.empty(x.shape[0], x.shape[1], x.device)
        out.hi.copy_(hi)
        out.lo.copy_(lo)
        return out

    @property
    def lo(self) -> torch.Tensor:
        return torch.as_strided(self.hi, self.hi.shape, self.hi.stride(), self.hi.storage_offset() + self.ld // 2)

    @property
    def shape(self):
        return self.hi.shape

    @property
    def device(self):
        return self.hi.device

    def dim(self):
        return 2

    def numel(self):
        return self.hi.numel()

    def data_ptr(self):
        return self.hi.data_ptr()

    def float(self) -> torch.Tensor:
        return self.hi.float() + self.lo.float()

    def cpu(self) -> torch.Tensor:
        return self.float().cpu()

    def view(self, *shape) -> torch.Tensor:
        """fp32 materialisation viewed as `shape` (reporting / tests: e.g. (B, S, d) encoder states)."""
        return self.float().view(*shape)

    def reshape(self, *shape) -> torch.Tensor:
        return self.float().reshape(*shape)

    def __getitem__(self, idx) -> "X2":
        """Row / column slices (step 1) keep the row stride, hence the lo-plane offset."""
        if not isinstance(idx, tuple):
            idx = (idx,)
        for s in idx:
            assert isinstance(s, slice) and s.step in (None, 1), "X2 views support plain slices only"
        return X2(self.hi[idx], self.ld)

    def record_stream(self, stream):
        self.hi.record_stream(stream)
