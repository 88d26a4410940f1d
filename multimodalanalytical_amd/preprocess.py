"""Device-side input path: the reference's `PatchPreprocessor`
(data/preprocessing/patches.py:14-107) with the same constructor fields, `initialise` and `__call__`
contract, running the standardise / interpolate / patchify / gradient / mask work as one HIP launch
(`afm_patch_preprocess`) on spectra already resident in HBM.  No CPU fallback.

`__call__` accepts what the reference accepts (a list of per-sample lists with `None` for a missing
spectrum) or a device tensor `(B, L)` plus a `present` mask, and returns `(patches, mask)` exactly as
the reference does: patches `(B, P, patch_size)` fp32, mask `(B, P)` bool with True = pad.  With
`seq_first=True` both come out sequence-first, the layout the collator puts into the batch dict
(datamodules.py:201-218), so no transpose pass is needed before the embedding.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import ops


@dataclass
class PatchPreprocessor:
    patch_size: int
    masking: bool
    interpolation: bool
    overlap: int = 1
    derivative: bool = False
    encoding_type: str = ""
    mean: float = field(init=False, default=0.0)
    std: float = field(init=False, default=1.0)
    device: str = "cuda:0"

    def initialise(self, sampled_dataset, modality: str) -> None:
        """Statistics over the NON-ZERO entries of the sampled spectra (patches.py:35-39)."""
        spectra = np.array(sampled_dataset[modality])
        nz = spectra[spectra != 0]
        self.mean = float(nz.mean())
        self.std = float(nz.std())

    def to_device(self, spectra: Sequence[Optional[Sequence[float]]]) -> Tuple[torch.Tensor, torch.Tensor]:
        """The reference's `None` handling (patches.py:63-67): missing rows become zeros of the longest
        present length (500 if every row is missing) and are flagged absent."""
        sizes = [len(s) if s is not None else -1 for s in spectra]
        L = max(sizes) if max(sizes) != -1 else 500
        host = np.zeros((len(spectra), L), dtype=np.float32)
        for i, s in enumerate(spectra):
            if s is not None:
                if len(s) != L:
                    raise ValueError("spectra of one batch must have one length (torch.Tensor(list) in the reference)")
                host[i] = np.asarray(s, dtype=np.float32)
        present = torch.tensor([n != -1 for n in sizes], dtype=torch.bool)
        return torch.from_numpy(host).to(self.device), present.to(self.device)

    def __call__(self, spectra: Union[torch.Tensor, List[Optional[List[float]]]],
                 present: Optional[torch.Tensor] = None, seq_first: bool = False):
        if not isinstance(spectra, torch.Tensor):
            spectra, present = self.to_device(spectra)
        return ops.patch_preprocess(spectra, present, self.mean, self.std, self.patch_size, masking=self.masking,
                                    interpolation=self.interpolation, overlap=self.overlap,
                                    derivative=self.derivative, seq_first=seq_first)


class DeviceCollator:
    """Batch-dict assembly of `MultiModalDataCollator.__call__` / `prepare_encoder_input`
    (data/datamodules.py:140-228, 230-351) for PRE-TOKENISED shards already in HBM: text-like modalities
    arrive as padded id matrices `(B, S_m)` int64 with their attention masks (1 = token), patch modalities
    as raw spectra `(B, L)` fp32 (+ `present`), the target as padded ids `(B, T+1)`.  Output: the
    reference's sequence-first batch dict (datamodules.py:201-218) --

        encoder_input{m}   text (S_m, B) int64 | patches (P, B, ps) fp32
        encoder_pad_mask   (sum S_m, B) bool, True = pad, modalities concatenated in input order
        decoder_input{t}   ids[:-1]   decoder_pad_mask  ~mask[:-1]
        target             ids[1:]    target_mask       ~mask[1:]

    Tokenisation itself (HF tokenizers / regex, data/tokenizer.py) stays on the host and out of scope;
    the spectrum work runs in `afm_patch_preprocess` and lands sequence-first without a transpose pass.
    """

    ALIGN_LEN = 1800     # datamodules.py:151-160: the alignment target is zero-padded to 1800 points

    def __init__(self, data_config: dict, preprocessors: dict, target_modality: str = None):
        self.data_config = data_config
        self.preprocessors = preprocessors          # {modality: PatchPreprocessor} for 1D_patches modalities
        # datamodules.py:39-66: inputs = not target; target = target and not alignment (exactly one); alignment = target and
        # alignment (at most one): its raw spectrum becomes `encoder_alignment_input`, the alignment head's regression target
        self.input_modalities = [m for m, c in data_config.items() if not c.get("target", False)]
        targets = [m for m, c in data_config.items() if c.get("target", False) and not c.get("alignment", False)]
        self.alignment_modality = [m for m, c in data_config.items() if c.get("target", False) and c.get("alignment", False)]
        if len(self.alignment_modality) > 1:
            raise ValueError("At most 1 target alignment modality can be specified.")
        if len(targets) != 1:
            raise ValueError("Only 1 target modality can be specified.")
        if target_modality is not None and target_modality != targets[0]:
            raise ValueError(f"target modality {target_modality!r} is not the data config's ({targets[0]!r})")
        self.target_modality = targets[0]

    def __call__(self, inputs: dict) -> dict:
        enc, masks = {}, []
        for m in self.input_modalities:
            cfg = self.data_config[m]
            x = inputs[m]
            if cfg["type"] == "1D_patches":
                spectra, present = (x["spectra"], x.get("present")) if isinstance(x, dict) else (x, None)
                patches, mask = self.preprocessors[m](spectra, present, seq_first=True)
                enc[m] = patches
                masks.append(mask)
            else:
                ids, att = x["input_ids"], x["attention_mask"]
                if "numerical_values" in x:
                    enc[m] = {"tokenized_input": ids.transpose(0, 1), "numerical_values": x["numerical_values"].transpose(0, 1)}
                else:
                    enc[m] = ids.transpose(0, 1)
                masks.append(~att.transpose(0, 1).bool())
        tgt = inputs[self.target_modality]
        ids = tgt["input_ids"].transpose(0, 1)
        pad = ~tgt["attention_mask"].transpose(0, 1).bool()
        out = {"encoder_input": enc, "encoder_pad_mask": torch.cat(masks, 0),
               "decoder_input": {self.target_modality: ids[:-1, :]}, "decoder_pad_mask": pad[:-1, :],
               "target": ids.clone()[1:, :], "target_mask": pad.clone()[1:, :]}
        if len(self.alignment_modality) == 1:       # datamodules.py:148-169, 211-212
            am = self.alignment_modality[0]
            B = ids.shape[1]
            if am not in inputs or inputs[am] is None:
                a = torch.zeros(B, self.ALIGN_LEN, dtype=torch.float32, device=ids.device)
            else:
                x = inputs[am]
                a = (x["spectra"] if isinstance(x, dict) else x).to(torch.float32)
            if a.shape[1] < self.ALIGN_LEN:
                a = torch.nn.functional.pad(a, (0, self.ALIGN_LEN - a.shape[1]), "constant", 0)
            pp = self.preprocessors.get(am)
            if self.data_config[am]["type"] == "1D_patches" and pp is not None and pp.interpolation:
                # the reference CALLS the boolean field here (`self.preprocessors[m].interpolation(alignment_input)`): it cannot run
                raise TypeError("'bool' object is not callable (reference datamodules.py:161-168: alignment targets with "
                                "preprocessor_arguments.interpolation = True fail there too)")
            out["encoder_alignment_input"] = a.contiguous()
        return out


def mix_indices(n_rows: int, mix_config: dict, split: str, seed: int = 3247):
    """Index stream of the reference's `mix_spectra` generator (data/datasets.py:59-116): numpy global RNG
    seeded once, `np.random.choice(range(n_rows), (parallel_samples, n_compounds))` per round, duplicates
    removed with `np.unique(axis=0)`, rows repeating a compound dropped, stop once
    `n * parallel + parallel >= perm(n_rows, n_compounds)`."""
    import math
    np.random.seed(seed)
    nc, par = mix_config["n_compounds"], mix_config["parallel_samples"]
    max_n = mix_config[f"{split}_max_n_samples"]
    if max_n // par < 1:
        par = max_n
    expected = math.perm(n_rows, nc)
    a = list(range(n_rows))
    for n in range(max_n // par):
        ri = np.unique(np.random.choice(a, size=(par, nc)), axis=0)
        ri = ri[np.array([len(set(row)) == len(row) for row in ri])]
        if n * par + par >= expected:
            break
        yield ri


def shard_rows(ri, rank: int, world_size: int):
    """Rank-strided shard of one round of the index stream (the reference's iterable mixture dataset has no
    DistributedSampler, so under DDP every rank would train on the SAME mixtures).  Every rank draws the identical
    stream (same seed) and keeps rows rank, rank + world, ...; the tail that does not divide evenly is dropped so
    all ranks run the same number of samples (and optimiser steps) per round."""
    if world_size <= 1:
        return ri
    n = (len(ri) // world_size) * world_size
    return ri[:n][rank::world_size]


class MixtureGenerator:
    """Device-side counterpart of `mix_spectra` (data/datasets.py:58-141) over a spectra table resident in
    HBM: the index stream stays on the host (numpy RNG, as the reference), the weighted average /
    normalisation / padding of every round runs in one `afm_mix_spectra` launch.  Each round yields

        indices      (n, n_compounds) int64   rows of the table that were mixed
        IR           (n * k, 1800) fp32        the mixed spectrum, repeated for each compound with ratio > 0
        compound     (n * k,) int64            table row whose Smiles / Formula is the target of that record
        IR_target    (n * k, L) fp32           the pure spectrum of that compound (the alignment target)
        Percentage   (n * k,) float64

    in the reference's record order (for idx in indices: for i in compounds)."""

    def __init__(self, table: torch.Tensor, mix_config: dict, split: str = "train", seed: int = 3247,
                 rank: int = 0, world_size: int = 1):
        self.table, self.cfg, self.split, self.seed = table, dict(mix_config), split, seed
        self.rank, self.world_size = int(rank), int(world_size)
        nc = self.cfg["n_compounds"]
        ratio = self.cfg.get("compounds_ratio") or [1 / nc] * nc
        if len(ratio) != nc or sum(ratio) != 1:
            raise ValueError(f"Invalid compound ratios: expected {nc} compounds with ratios summing to 1.")
        self.ratio = list(ratio)

    def __iter__(self):
        dev = self.table.device
        keep = [i for i, r in enumerate(self.ratio) if r != 0]
        if self.cfg.get("mixed", False):
            # data/datasets.py:92-105: the table already holds mixtures; every row goes out once, normalised if asked, with a
            # mock target (equal ratios only, as the reference insists)
            nc = self.cfg["n_compounds"]
            if self.ratio != [1 / nc] * nc:
                raise ValueError("Mixed mode is only supported with equal compound ratios at the moment.")
            rows = shard_rows(np.arange(self.table.shape[0], dtype=np.int64)[:, None], self.rank, self.world_size)
            idx = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
            L = self.table.shape[1]
            ir = (ops.mix_spectra(self.table, idx, [1.0], normalize=True, out_len=L) if self.cfg.get("normalize", False)
                  else self.table[idx[:, 0]])
            yield {"indices": rows, "IR": ir, "compound": idx[:, 0], "IR_target": torch.zeros_like(ir),
                   "Percentage": torch.full((len(rows),), 1 / nc, dtype=torch.float64)}
            return
        for ri in mix_indices(self.table.shape[0], self.cfg, self.split, self.seed):
            ri = shard_rows(ri, self.rank, self.world_size)
            if len(ri) == 0:
                continue
            idx = torch.from_numpy(np.ascontiguousarray(ri)).to(dev)
            mixed = ops.mix_spectra(self.table, idx, self.ratio, normalize=bool(self.cfg.get("normalize", False)))
            comp = idx[:, keep].reshape(-1)
            yield {"indices": ri, "IR": mixed.repeat_interleave(len(keep), dim=0), "compound": comp,
                   "IR_target": self.table[comp],
                   "Percentage": torch.tensor([self.ratio[i] for i in keep] * len(ri), dtype=torch.float64)}


def interleave_rounds(streams):
    """`multi_config_mix` (data/datasets.py:24-46): the record streams of several mixture configurations alternate record by record
    (zip_longest: an exhausted stream drops out, the others go on).  `streams`: iterables of dicts of equally long tensors (the rounds
    of a MixtureGenerator); yields dicts in the interleaved order, at most one round's worth per yield."""
    its = [iter(s) for s in streams]
    if len(its) == 1:
        for r in its[0]:
            yield {k: v for k, v in r.items() if k != "indices"}
        return
    buf = [None] * len(its)          # unread tail of each stream's current round
    live = [True] * len(its)

    def n_rows(d):
        return int(next(iter(d.values())).shape[0])

    def refill(i):
        while live[i] and (buf[i] is None or n_rows(buf[i]) == 0):
            try:
                r = next(its[i])
                buf[i] = {k: v for k, v in r.items() if k != "indices"}
            except StopIteration:
                live[i], buf[i] = False, None
    while True:
        for i in range(len(its)):
            refill(i)
        act = [i for i in range(len(its)) if live[i]]
        if not act:
            return
        n = min(n_rows(buf[i]) for i in act)                       # records every live stream can contribute right now
        out = {}
        for k in buf[act[0]]:
            parts = [buf[i][k][:n] for i in act]
            if torch.is_tensor(parts[0]):
                out[k] = torch.stack(parts, 1).reshape((n * len(act),) + tuple(parts[0].shape[1:]))
            else:
                out[k] = np.stack(parts, 1).reshape((n * len(act),) + tuple(parts[0].shape[1:]))
        for i in act:
            buf[i] = {k: v[n:] for k, v in buf[i].items()}
        yield out
