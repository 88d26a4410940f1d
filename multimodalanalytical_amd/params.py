"""Flat parameter store of the spectra->SMILES model.

Every tensor of the reference's `CustomModel` state dict (custom_modeling.py:323-418,
modeling/utils.py:44-82; key layout in SURVEY.md section 5.4) is a VIEW of one flat fp32
buffer in HBM; gradients, Adam moments and the bf16 shadow used by the MFMA GEMMs are flat
buffers with the same offsets.  One buffer means: one fused clip+Adam launch, one
contiguous range per gradient all-reduce bucket, and checkpoint keys identical to the
reference's.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Any, Dict, List, Tuple

import torch

TEXT_TYPES = ("text", "text_spectrum", "peak_positional_encoding", "run_length_encoding",
              "multiplets", "carbon", "msms_text")          # modeling/utils.py:93-101
PATCH_TYPES = ("1D_patches", "msms_number")                  # modeling/utils.py:107

ALIGN = 8  # elements: keeps fp32 views 32-B and bf16 shadow views 16-B aligned


class Spec:
    __slots__ = ("name", "shape", "kind", "offset", "numel", "fan_in")

    def __init__(self, name, shape, kind, fan_in=None):
        self.name, self.shape, self.kind, self.fan_in = name, tuple(shape), kind, fan_in
        self.numel = int(math.prod(shape))
        self.offset = -1


def patch_layers(mcfg: Dict[str, Any], d: int) -> List[Tuple[str, int, int]]:
    """(key suffix, in, out) of the Linear stack of a patch embedder (modeling/utils.py:107-136)."""
    if mcfg["type"] == "msms_number":
        ps = 2
    else:
        ps = int(mcfg["preprocessor_arguments"]["patch_size"])
    enc = (mcfg.get("preprocessor_arguments") or {}).get("encoding_type", "linear")
    if enc == "linear":
        return [("", ps, d)]
    if enc == "linear_2_layer":
        return [("0.", ps, d // 2), ("2.", d // 2, d)]
    if enc == "linear_3_layer":
        return [("0.", ps, d // 3), ("2.", d // 3, 2 * (d // 3)), ("4.", 2 * (d // 3), d)]
    raise NotImplementedError(enc)


def build_specs(cfg: Dict[str, Any], data_config: Dict[str, Any], vocab_out: int) -> List[Spec]:
    d = cfg["d_model"]
    sp: List[Spec] = []

    def lin(prefix, out, inp):
        sp.append(Spec(prefix + "weight", (out, inp), "matrix"))
        sp.append(Spec(prefix + "bias", (out,), "linear_bias", fan_in=inp))

    def ln(prefix):
        sp.append(Spec(prefix + "weight", (d,), "ones"))
        sp.append(Spec(prefix + "bias", (d,), "zeros"))

    for m, mc in data_config.items():
        p = f"embedding.embedding_layer_dict.{m}."
        if mc["type"] in TEXT_TYPES:
            sp.append(Spec(p + "weight", (int(mc["vocab_size"]), d), "matrix"))
        elif mc["type"] in PATCH_TYPES:
            for suf, i, o in patch_layers(mc, d):
                lin(p + suf, o, i)
        else:
            raise NotImplementedError(f"Unknown modality type: {mc['type']}")
    if cfg.get("multimodal_norm", True):
        for m in data_config:
            ln(f"embedding.embedding_norm_dict.{m}.")
    if cfg["positional_encoding_type"] == "learned":
        sp.append(Spec("embedding.positional_encodings.pos_encodings.weight",
                       (cfg["max_position_embeddings"], d), "matrix"))
        ln("embedding.positional_encodings.norm.")
    elif cfg["positional_encoding_type"] != "sin_cos":
        raise KeyError(cfg["positional_encoding_type"])

    def attn(prefix):
        sp.append(Spec(prefix + "in_proj_weight", (3 * d, d), "matrix"))
        sp.append(Spec(prefix + "in_proj_bias", (3 * d,), "zeros"))
        sp.append(Spec(prefix + "out_proj.weight", (d, d), "matrix"))
        sp.append(Spec(prefix + "out_proj.bias", (d,), "zeros"))

    def ffn(prefix, f):
        # linear1 and gate are adjacent so the gated FFN runs ONE (2f x d) GEMM
        sp.append(Spec(prefix + "linear1.weight", (f, d), "matrix"))
        if cfg["gated_linear"]:
            sp.append(Spec(prefix + "gate.weight", (f, d), "matrix"))
        sp.append(Spec(prefix + "linear1.bias", (f,), "linear_bias", fan_in=d))
        if cfg["gated_linear"]:
            sp.append(Spec(prefix + "gate.bias", (f,), "linear_bias", fan_in=d))
        lin(prefix + "linear2.", d, f)

    for i in range(cfg["encoder_layers"]):
        p = f"encoder.layers.{i}."
        attn(p + "self_attn.")
        ffn(p, cfg["encoder_ffn_dim"])
        ln(p + "norm1."); ln(p + "norm2.")
    ln("encoder.norm.")
    for i in range(cfg["decoder_layers"]):
        p = f"decoder.layers.{i}."
        attn(p + "self_attn.")
        attn(p + "multihead_attn.")
        ffn(p, cfg["decoder_ffn_dim"])
        ln(p + "norm1."); ln(p + "norm2."); ln(p + "norm3.")
    ln("decoder.norm.")
    lin("token_ff.", vocab_out, d)
    # encoder alignment head (custom_modeling.py:363-396), LAST in the flat buffer: its backward runs
    # first, so "gradients from offset X to the end are final" keeps holding for the DDP bucket hooks
    ac = align_dict(cfg.get("align_config"))
    if ac:
        hid, n_out = int(ac["hidden_dimension"]), int(ac["output_dimension"])
        lin("align_network.0.", hid, d)
        if ac["align_network"] == "mlp":
            lin("align_network.2.", n_out, hid)
        elif ac["align_network"] == "convolutional":
            C, k = int(ac["conv_channels"]), int(ac["kernel_size"])
            lin("align_network.2.", hid, hid)
            sp.append(Spec("align_network.4.weight", (C, hid, k), "matrix"))
            sp.append(Spec("align_network.4.bias", (C,), "linear_bias", fan_in=hid * k))
            sp.append(Spec("align_network.6.weight", (n_out, C, 1), "matrix"))
            sp.append(Spec("align_network.6.bias", (n_out,), "linear_bias", fan_in=C))
        else:
            raise ValueError(f"unknown align_network {ac['align_network']}")
    return sp


def align_dict(ac):
    """AlignConfig object (custom_modeling.py:18-37) or yaml dict -> plain dict (None if absent)."""
    if not ac:
        return None
    return dict(ac) if isinstance(ac, dict) else dict(vars(ac))


class ParamStore:
    def __init__(self, specs: List[Spec], device, with_bf16: bool, with_x2: bool = False, lowp_dtype=torch.bfloat16):
        off = 0
        self.specs: "OrderedDict[str, Spec]" = OrderedDict()
        for s in specs:
            s.offset = off
            off += (s.numel + ALIGN - 1) // ALIGN * ALIGN
            self.specs[s.name] = s
        self.total = off
        self.device = device
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=device)
        # flat 16-bit shadow of every parameter (single-pass modes: bf16 or fp16), written by the Adam kernel
        self.bf16 = torch.zeros(off, dtype=lowp_dtype, device=device) if with_bf16 else None
        # split-pair shadow of the GEMM weights (bf16x3 mode): the (rows x cols) matrix at flat offset o lives at
        # [2 o, 2 o + 2 rows cols) as rows of [hi(cols) | lo(cols)]
        self.x2buf = torch.zeros(2 * off, dtype=torch.bfloat16, device=device) if with_x2 else None
        self.num_params = sum(s.numel for s in specs)
        self._span_ok = set()

    # ---- views
    def _view(self, buf, name):
        s = self.specs[name]
        return buf[s.offset:s.offset + s.numel].view(s.shape)

    def p(self, name):
        return self._view(self.flat, name)

    def g(self, name):
        return self._view(self.grad, name)

    def pb(self, name):
        return self._view(self.bf16, name)

    def span(self, buf, first: str, rows: int, cols: int):
        """(rows x cols) view starting at `first` that may cover several adjacent tensors."""
        s = self.specs[first]
        self._check_span(first, rows * cols)
        return buf[s.offset:s.offset + rows * cols].view(rows, cols)

    def span_x2(self, first: str, rows: int, cols: int):
        """Split-pair (rows x cols) weight view starting at `first` (see x2buf)."""
        from .x2 import X2
        s = self.specs[first]
        self._check_span(first, rows * cols)
        phys = self.x2buf[2 * s.offset:2 * s.offset + 2 * rows * cols].view(rows, 2 * cols)
        return X2(phys[:, :cols], 2 * cols)

    def vec_span(self, buf, first: str, i0: int, i1: int):
        """Elements [i0, i1) of the vector starting at `first` (may run into adjacent tensors: packed biases)."""
        s = self.specs[first]
        if i1 > s.numel:
            self._check_span(first, i1)
        return buf[s.offset + i0:s.offset + i1]

    def _check_span(self, first: str, numel: int) -> None:
        """The covered tensors must tile [offset, offset + numel) exactly: every spec is padded to ALIGN
        elements, so a tensor whose numel is not a multiple of ALIGN in the middle of a span (e.g. a gated FFN
        with ffn_dim % 8 != 0) would shift its neighbours inside the view."""
        key = (first, numel)
        if key in self._span_ok:
            return
        names = list(self.specs)
        i, covered, expect = names.index(first), 0, self.specs[first].offset
        while covered < numel:
            sp = self.specs[names[i]]
            if sp.offset != expect or covered + sp.numel > numel and sp.numel != numel - covered:
                raise ValueError(f"span({first}, {numel}): {sp.name} does not continue the view contiguously "
                                 f"(dimensions must be multiples of {ALIGN})")
            covered += sp.numel
            expect = sp.offset + sp.numel     # the NEXT tensor must start right here, i.e. no padding in between
            i += 1
        self._span_ok.add(key)

    def names(self):
        return list(self.specs)

    # ---- init (HFWrapper._init_params + torch module defaults, wrapper.py:320-327)
    def init_(self, seed: int) -> None:
        g = torch.Generator().manual_seed(int(seed))
        host = torch.zeros(self.total, dtype=torch.float32)
        for s in self.specs.values():
            v = host[s.offset:s.offset + s.numel].view(s.shape)
            if s.kind == "matrix":        # xavier_uniform_ on every parameter with dim > 1
                rf = int(math.prod(s.shape[2:]))            # Conv1d weights: receptive field = kernel size
                fan_out, fan_in = s.shape[0] * rf, s.shape[1] * rf
                a = math.sqrt(6.0 / (fan_in + fan_out))
                v.copy_((torch.rand(s.shape, generator=g) * 2 - 1) * a)
            elif s.kind == "linear_bias":  # nn.Linear default: U(+-1/sqrt(fan_in))
                a = 1.0 / math.sqrt(s.fan_in)
                v.copy_((torch.rand(s.shape, generator=g) * 2 - 1) * a)
            elif s.kind == "ones":
                v.fill_(1.0)
        self.flat.copy_(host)

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((n, self.p(n).detach().clone()) for n in self.specs)

    def load(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        missing = [n for n in self.specs if n not in sd]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        host = self.flat.detach().cpu()
        for n, s in self.specs.items():
            if n in sd:
                t = sd[n].detach().to(torch.float32).cpu()
                if tuple(t.shape) != s.shape:
                    raise ValueError(f"{n}: shape {tuple(t.shape)} != {s.shape}")
                host[s.offset:s.offset + s.numel].view(s.shape).copy_(t)
        self.flat.copy_(host)
