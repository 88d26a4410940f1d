"""CPU: the C-ABI library builds/loads and exports every symbol include/afm_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "afm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(afm_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from multimodalanalytical_amd import lib as L
    handle = L.load()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(handle, name), f"libafm_hip.so does not export {name}"
    assert sorted(L.exported_symbols()) == declared, "lib.py binding and header disagree"
    assert handle.afm_abi_version() == L.ABI_VERSION == 6
    for which, st in enumerate((L.Dropout, L.GemmDesc, L.LnShape, L.AttnShape, L.PatchDesc, L.BeamDesc, L.CastItem)):
        assert handle.afm_struct_size(which) == ctypes.sizeof(st)
    assert handle.afm_error_string(-2) == b"unsupported shape/dtype for the requested algorithm"


def test_struct_layouts_match_the_header():
    from multimodalanalytical_amd import lib as L
    # 11 int32 (+pad) + 7 pointers + 4 int32 + dropout{float,u32,u64} + glu_rows, reserved2 + k_live
    assert ctypes.sizeof(L.Dropout) == 16
    assert ctypes.sizeof(L.GemmDesc) == 48 + 7 * 8 + 16 + 16 + 8 + 8 and L.GemmDesc.glu_rows.offset == 136 and L.GemmDesc.k_live.offset == 144
    assert L.GemmDesc.A.offset == 48 and L.GemmDesc.a_colsum.offset == 96 and L.GemmDesc.drop.offset == 120
    assert ctypes.sizeof(L.LnShape) == 88 and L.LnShape.add_drop.offset == 48 and L.LnShape.row_live.offset == 64 and L.LnShape.flags.offset == 72 and L.LnShape.row_map.offset == 80
    assert L.AttnShape.key_pad.offset == 56 and L.AttnShape.sqb.offset == 80 and ctypes.sizeof(L.AttnShape) == 136
    assert L.PatchDesc.mean.offset == 32 and ctypes.sizeof(L.PatchDesc) == 48


def test_integration_md_struct_example_matches_the_library():
    """The ctypes structs INTEGRATION.md shows a maintainer are executed as written and held to the library's own sizes
    (the example had lost `k_live`, 8 bytes short of the ABI-4 descriptor)."""
    from multimodalanalytical_amd import lib as L
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    block = next(b for b in blocks if "class AfmGemmDesc" in b)
    classes = block[block.index("class AfmDropout"):block.index("lib = ctypes.CDLL")]
    ns = {"ctypes": ctypes}
    exec(compile(classes, "INTEGRATION.md", "exec"), ns)
    handle = L.load()
    assert ctypes.sizeof(ns["AfmDropout"]) == handle.afm_struct_size(0)
    assert ctypes.sizeof(ns["AfmGemmDesc"]) == handle.afm_struct_size(1)
    assert [f[0] for f in ns["AfmGemmDesc"]._fields_] == [f[0] for f in L.GemmDesc._fields_]
    for name, _ in L.GemmDesc._fields_:
        assert getattr(ns["AfmGemmDesc"], name).offset == getattr(L.GemmDesc, name).offset, name
    assert "afm_abi_version() == %d" % L.ABI_VERSION in text


def test_patch_count_is_host_side():
    """afm_patch_count is pure host arithmetic: usable here (sizes of PatchPreprocessor outputs, patches.py:79-96)."""
    from multimodalanalytical_amd import lib as L
    h = L.load()
    def P(**kw):
        base = dict(B=4, L=1800, patch_size=125, step=125, interpolation=0, derivative=0, masking=0, seq_first=0, mean=0.0, std=1.0)
        base.update(kw)
        d = L.PatchDesc(**base)
        return h.afm_patch_count(ctypes.byref(d))
    assert P() == 14 and P(patch_size=75, step=75, interpolation=1) == 21 and P(L=1984, patch_size=2, step=2) == 992
    assert P(patch_size=50, step=25) == 71 and P(patch_size=100, step=100, derivative=1) == 36
    assert P(std=0.0) == -1 and P(L=1000, interpolation=1) == -1 and P(patch_size=0) == -1


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from multimodalanalytical_amd import ops
    from multimodalanalytical_amd.lib import AfmError
    a = torch.zeros(4, 8)
    with pytest.raises(AfmError):
        ops.gemm(a, a, torch.zeros(4, 4))   # the product path has no CPU fallback
