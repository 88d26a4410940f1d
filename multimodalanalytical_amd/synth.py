"""Synthetic "synth-177K" batches (SURVEY.md section 8d / BASELINE.md section 2): seeded, generated
on the box, no chemistry.  Produces the SAME seq-first batch dict the reference's
MultiModalDataCollator emits (reference data/datamodules.py:201-218): text modalities (S_m, B)
int64, patch modalities (P, B, ps) fp32, `encoder_pad_mask` (S, B) bool True = pad,
`decoder_input` / `target` shifted by one, `decoder_pad_mask`.

Workloads (BASELINE.json `configs`):
  c1  tiny 2L/d128, IR-only (Formula 12 + 14 patches of 125), T=40, B=8       (plumbing)
  c2  base 6L/d512/h8/f2048, IR-only: Formula 32 + 992 patches of 2 -> S=1024, T=128
  c3  base, Formula 32 + IR 24x75 + Multiplets 968 -> S=1024
  c4  12L/d768/h12/f3072, Formula 32 + IR 24x75 + Multiplets 768 + Carbon 200 -> S=1024
  c5  base gated, Formula 32 + IR 24x75 -> S=56, T=256 (mixture decode shape)
"""
from __future__ import annotations

from typing import Any, Dict, Tuple

import numpy as np
import torch

PAD, UNK, BOS, EOS = 0, 1, 2, 3  # data/tokenizer.py:22


def _text(vocab, target=False):
    return {"type": "text", "vocab_size": vocab, "pad_token_id": PAD, "target": target}


def _patch(ps):
    return {"type": "1D_patches", "target": False,
            "preprocessor_arguments": {"patch_size": ps, "interpolation": False, "masking": False}}


BASE = dict(d_model=512, encoder_layers=6, decoder_layers=6, encoder_attention_heads=8, decoder_attention_heads=8,
            encoder_ffn_dim=2048, decoder_ffn_dim=2048, dropout=0.1, gated_linear=False,
            positional_encoding_type="sin_cos", max_position_embeddings=1024, multimodal_norm=True)

WORKLOADS: Dict[str, Dict[str, Any]] = {
    "c1": dict(cfg=dict(BASE, d_model=128, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=512, decoder_ffn_dim=512),
               data={"Formula": _text(64), "IR": _patch(125), "Smiles": _text(128, True)},
               lens={"Formula": (12, 4, 10), "IR": 14}, T=40, batch=8),
    "c2": dict(cfg=dict(BASE), data={"Formula": _text(64), "IR": _patch(2), "Smiles": _text(128, True)},
               lens={"Formula": (32, 6, 20), "IR": 992}, T=128, batch=128),
    "c3": dict(cfg=dict(BASE), data={"Formula": _text(64), "IR": _patch(75),
                                     "Multiplets": {**_text(2048), "type": "multiplets"}, "Smiles": _text(128, True)},
               lens={"Formula": (32, 6, 20), "IR": 24, "Multiplets": (968, 100, 760)}, T=128, batch=128),
    "c4": dict(cfg=dict(BASE, d_model=768, encoder_layers=12, decoder_layers=12, encoder_attention_heads=12,
                        decoder_attention_heads=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072, gated_linear=True,
                        positional_encoding_type="learned"),
               data={"Formula": _text(64), "IR": _patch(75), "Multiplets": {**_text(2048), "type": "multiplets"},
                     "Carbon": {**_text(2304), "type": "carbon"}, "Smiles": _text(128, True)},
               lens={"Formula": (32, 6, 20), "IR": 24, "Multiplets": (768, 100, 760), "Carbon": (200, 20, 190)},
               T=128, batch=128),
    # mixture IR -> SMILES with the alignment head of the reference's mixture runs (configs/model/custom_model_align.yaml:29-36,
    # data/ir/patches_mixture_text_align.yaml): IR = the mean of two pure spectra, IR_target = the pure spectrum of the target compound
    "c5": dict(cfg=dict(BASE, gated_linear=True,
                        align_config=dict(align_network="convolutional", hidden_dimension=256, conv_channels=512, kernel_size=5,
                                          output_dimension=1800, loss_lambda=50, loss_function="mae")),
               data={"Formula": _text(64), "IR": _patch(75),
                     "IR_target": {**_patch(75), "target": True, "alignment": True}, "Smiles": _text(128, True)},
               lens={"Formula": (32, 6, 20), "IR": 24}, T=256, batch=128),
}


def train_flops_per_sample(cfg, S, T, V, patch_flops=0.0) -> float:
    """3 x forward FLOPs: Le(8Sd^2+4S^2d+gSdf) + Ld(12Td^2+4Sd^2+4T^2d+4TSd+gTdf) + 2TdV."""
    d, g = cfg["d_model"], (6 if cfg["gated_linear"] else 4)
    fe, fd = cfg["encoder_ffn_dim"], cfg["decoder_ffn_dim"]
    enc = cfg["encoder_layers"] * (8 * S * d * d + 4 * S * S * d + g * S * d * fe)
    dec = cfg["decoder_layers"] * (12 * T * d * d + 4 * S * d * d + 4 * T * T * d + 4 * T * S * d + g * T * d * fd)
    return 3.0 * (enc + dec + 2 * T * d * V + patch_flops)


def executed_flops_per_sample(cfg, S, T, V, enc_live=None, enc_live_sq=None, fused_cross=None) -> Dict[str, float]:
    """What the kernels really multiply per trained sample (bench.py `step_mfma_frac_executed`, `live_gflop_per_sample`).
      executed  the algorithmic count + the products the attention BACKWARD recomputes: the two backward kernels evaluate 7 products
                of S x S x dh per head where the algorithm has 4 (Q K^T and dO V^T once more in each of them minus the shared one),
                so an attention instance costs 9 products per layer instead of 6.
      live      the same over the LIVE encoder positions only: the work a training step cannot leave out.  Since round 6 both
                directions leave padded rows out (the forward in whole 256-row groups after per-sample compaction, the
                backward in 64-row blocks / 256-row tiles), so what the kernels execute lies between `live` and `executed`.
                enc_live = mean live encoder length, enc_live_sq = mean of its square.
      fused_cross  the decoder's cross-attention backward runs as ONE kernel (round 6: 5 products, one recomputed, instead of 7;
                default: where the engine's default takes that kernel, T <= 128)."""
    if fused_cross is None:
        fused_cross = T <= 128
    xb = 5 if fused_cross else 7          # products of a cross-attention instance's backward
    d, g = cfg["d_model"], (6 if cfg["gated_linear"] else 4)
    fe, fd, Le, Ld = cfg["encoder_ffn_dim"], cfg["decoder_ffn_dim"], cfg["encoder_layers"], cfg["decoder_layers"]
    alg = train_flops_per_sample(cfg, S, T, V)
    extra = Le * 3 * 2 * S * S * d + Ld * (3 * 2 * T * T * d + (xb - 4) * 2 * T * S * d)       # recomputed products per attention instance: 3 (two kernels), 1 (fused cross)
    out = {"algorithmic": alg, "executed": alg + extra}
    if enc_live is not None:
        s1, s2 = float(enc_live), float(enc_live_sq if enc_live_sq is not None else enc_live * enc_live)
        fwd = Le * (8 * s1 * d * d + 4 * s2 * d + g * s1 * d * fe) + \
            Ld * (12 * T * d * d + 4 * s1 * d * d + 4 * T * T * d + 4 * T * s1 * d + g * T * d * fd) + 2 * T * d * V
        bwd_gemm = 2 * (Le * (8 * s1 * d * d + g * s1 * d * fe) + Ld * (12 * T * d * d + 4 * s1 * d * d + g * T * d * fd) + 2 * T * d * V)
        bwd_attn = Le * 7 * 2 * s2 * d + Ld * (7 * 2 * T * T * d + xb * 2 * T * s1 * d)
        out["live"] = fwd + bwd_gemm + bwd_attn
    return out


def make_batch(name: str, batch: int, seed: int, device="cpu") -> Tuple[Dict[str, Any], Dict[str, Any]]:
    """One seq-first batch dict of workload `name` (+ the workload record)."""
    w = WORKLOADS[name]
    rng = np.random.default_rng(seed)
    enc, masks = {}, []
    align = None
    for m, spec in w["lens"].items():
        mc = w["data"][m]
        if isinstance(spec, tuple):
            L, lo, hi = spec
            V = mc["vocab_size"]
            n = rng.integers(lo, hi + 1, size=batch) + 2          # + bos/eos
            ids = rng.integers(4, V, size=(L, batch)).astype(np.int64)
            pos = np.arange(L)[:, None]
            ids[0, :] = BOS
            ids[np.minimum(n - 1, L - 1), np.arange(batch)] = EOS
            pad = pos >= n[None, :]
            ids[pad] = PAD
            enc[m] = torch.from_numpy(ids)
            masks.append(torch.from_numpy(pad))
        else:
            P, ps = spec, mc["preprocessor_arguments"]["patch_size"]
            k = np.exp(-0.5 * (np.arange(-9, 10) / 3.0) ** 2); k /= k.sum()       # sigma=3 smoothing
            smooth = lambda a: np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, a)
            x = smooth(np.abs(rng.standard_normal((batch, P * ps))).astype(np.float32))
            if any(c.get("alignment") for c in w["data"].values()):
                # a mixture workload: the input is the mean of this compound's spectrum and a second one (mix_spectra, equal
                # ratios); the compound's own raw spectrum is the alignment head's target (datamodules.py:148-160)
                align = x.astype(np.float32)
                x = 0.5 * (x + smooth(np.abs(rng.standard_normal((batch, P * ps))).astype(np.float32)))
            x = (x - x.mean()) / (x.std() + 1e-8)                                  # PatchPreprocessor standardise
            enc[m] = torch.from_numpy(np.ascontiguousarray(x.reshape(batch, P, ps).transpose(1, 0, 2)).astype(np.float32))
            masks.append(torch.zeros(P, batch, dtype=torch.bool))
    T, V = w["T"], w["data"]["Smiles"]["vocab_size"]
    n = rng.integers(20, min(120, T - 8) + 1, size=batch) + 2
    ids = rng.integers(4, V, size=(T + 1, batch)).astype(np.int64)
    ids[0, :] = BOS
    ids[n - 1, np.arange(batch)] = EOS
    ids[np.arange(T + 1)[:, None] >= n[None, :]] = PAD
    ids = torch.from_numpy(ids)
    b = {
        "encoder_input": enc,
        "encoder_pad_mask": torch.cat(masks, 0),
        "decoder_input": {"Smiles": ids[:-1].contiguous()},
        "decoder_pad_mask": ids[:-1] == PAD,
        "target": ids[1:].contiguous(),
    }
    if align is not None:
        a = torch.from_numpy(align)
        b["encoder_alignment_input"] = torch.nn.functional.pad(a, (0, max(0, 1800 - a.shape[1])))
    if device != "cpu":
        b = to_device(b, device)
    return b, w


def to_device(b, device):
    if isinstance(b, dict):
        return {k: to_device(v, device) for k, v in b.items()}
    if torch.is_tensor(b):
        return b.to(device, non_blocking=True)
    return b


def make_shard(name: str, n: int = 177000, seed: int = 3247, device="cpu") -> Dict[str, Any]:
    """The synthetic set of SURVEY 8(d) ("synth-177K": 177 000 samples, seed 3247) for workload `name`, as ONE shard in the format
    `cli/training.py` reads (pre-tokenised ids + RAW spectra): what `bench.py` keeps resident in HBM and draws every micro-batch
    from through ShardLoader -> DeviceCollator -> afm_patch_preprocess inside the timed region.  Same layout and length
    distributions as `make_batch`; the spectra are raw (positive, smoothed), standardisation happens in the collator's
    PatchPreprocessor (statistics over the non-zero entries of the first 10 000 rows, patches.py:35-39)."""
    w = WORKLOADS[name]
    g = torch.Generator(device=device).manual_seed(seed)
    data, meta = {}, {}

    def ids(L, lo, hi, V, first_real=4):
        nn = torch.randint(lo, hi + 1, (n,), generator=g, device=device) + 2          # + bos / eos
        x = torch.randint(first_real, V, (n, L), generator=g, device=device)
        pos = torch.arange(L, device=device)[None, :]
        x[:, 0] = BOS
        x.scatter_(1, torch.clamp(nn - 1, max=L - 1)[:, None], EOS)
        pad = pos >= nn[:, None]
        x[pad] = PAD
        return {"input_ids": x, "attention_mask": ~pad}

    for m, spec in w["lens"].items():
        mc = w["data"][m]
        if isinstance(spec, tuple):
            L, lo, hi = spec
            data[m] = ids(L, lo, hi, mc["vocab_size"])
            meta[m] = {"vocab_size": mc["vocab_size"], "pad_token_id": PAD}
        else:
            P, ps = spec, mc["preprocessor_arguments"]["patch_size"]
            k = torch.exp(-0.5 * (torch.arange(-9, 10, dtype=torch.float32) / 3.0) ** 2)
            k = (k / k.sum()).tolist()
            Lsp = P * ps
            out = torch.empty(n, Lsp, dtype=torch.float32, device=device)
            for i in range(0, n, 16384):                                               # sigma = 3 smoothing, in slabs
                x = torch.nn.functional.pad(torch.randn(min(16384, n - i), Lsp, generator=g, device=device).abs(), (9, 9))
                acc = torch.full((x.shape[0], Lsp), 0.05, dtype=torch.float32, device=device)
                for j, wj in enumerate(k):        # 19 shifted adds (plain elementwise kernels: no convolution library in the data path)
                    acc.add_(x[:, j:j + Lsp], alpha=wj)
                out[i:i + x.shape[0]] = acc
            data[m] = {"spectra": out}
    T, V = w["T"], w["data"]["Smiles"]["vocab_size"]
    data["Smiles"] = ids(T + 1, 20, min(120, T - 8), V)
    meta["Smiles"] = {"vocab_size": V, "pad_token_id": PAD}
    return {"meta": meta, "data": data}


def shard_collator(name: str, shard, device):
    """DeviceCollator of workload `name` over `shard` (PatchPreprocessor statistics from its first 10 000 spectra)."""
    from .preprocess import DeviceCollator, PatchPreprocessor
    w = WORKLOADS[name]
    dc = {m: c for m, c in w["data"].items() if not c.get("alignment")}
    pre = {}
    for m, mc in dc.items():
        if mc["type"] == "1D_patches":
            a = mc["preprocessor_arguments"]
            pp = PatchPreprocessor(patch_size=int(a["patch_size"]), masking=bool(a.get("masking", False)),
                                   interpolation=bool(a.get("interpolation", False)), device=str(device))
            pp.initialise({m: shard["data"][m]["spectra"][:10000].cpu().numpy()}, m)
            pre[m] = pp
    return DeviceCollator(dc, pre)


def synth_shards(data_config, n_train: int, n_val: int, n_test: int, seed: int = 3247, spectrum_len: int = 1800,
                 text_len: int = 32, target_len: int = 64, vocab: int = 64):
    """Pre-tokenised synthetic shards in the format `cli/training.py` reads, for ANY composed data config: text-like
    modalities as padded id matrices (bos ... eos pad), patch modalities as raw positive spectra."""
    rng = np.random.default_rng(seed)

    def ids(n, L, V):
        lens = rng.integers(max(3, L // 4), L - 1, size=n) + 1
        x = rng.integers(4, V, size=(n, L)).astype(np.int64)
        x[:, 0] = BOS
        x[np.arange(n), lens - 1] = EOS
        pad = np.arange(L)[None, :] >= lens[:, None]
        x[pad] = PAD
        return {"input_ids": torch.from_numpy(x), "attention_mask": torch.from_numpy(~pad)}

    def shard(n):
        data, meta = {}, {}
        for m, mc in data_config.items():
            if mc.get("alignment"):
                continue            # the alignment target is made by the mixture generator (or absent: the collator fills zeros)
            if mc["type"] == "1D_patches":
                x = np.abs(rng.standard_normal((n, spectrum_len))).astype(np.float32)
                k = np.exp(-0.5 * (np.arange(-9, 10) / 3.0) ** 2); k /= k.sum()
                x = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, x).astype(np.float32) + 0.05
                data[m] = {"spectra": torch.from_numpy(x)}
            else:
                L = target_len + 1 if mc.get("target") else text_len
                data[m] = ids(n, L, vocab)
                meta[m] = {"vocab_size": vocab, "pad_token_id": PAD}
        return {"meta": meta, "data": data}
    return {"train": shard(n_train), "val": shard(n_val), "test": shard(n_test)}


def write_shards(path: str, data_config, n_train: int, n_val: int, n_test: int, seed: int = 3247, **kw) -> None:
    import os
    os.makedirs(path, exist_ok=True)
    for split, sh in synth_shards(data_config, n_train, n_val, n_test, seed, **kw).items():
        torch.save(sh, os.path.join(path, split + ".pt"))
