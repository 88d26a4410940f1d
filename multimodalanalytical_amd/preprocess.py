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

    def __init__(self, data_config: dict, preprocessors: dict, target_modality: str):
        self.data_config = data_config
        self.preprocessors = preprocessors          # {modality: PatchPreprocessor} for 1D_patches modalities
        self.target_modality = target_modality
        self.input_modalities = [m for m, c in data_config.items() if not c.get("target", False)]

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
        return {"encoder_input": enc, "encoder_pad_mask": torch.cat(masks, 0),
                "decoder_input": {self.target_modality: ids[:-1, :]}, "decoder_pad_mask": pad[:-1, :],
                "target": ids.clone()[1:, :], "target_mask": pad.clone()[1:, :]}
