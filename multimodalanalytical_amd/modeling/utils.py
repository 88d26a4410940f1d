"""Host-side mirror of the reference's `analytical_fm.modeling.utils` surface
(reference modeling/utils.py): MultimodalEmbedding, POS_ENC_REGISTRY, CustomLMOutput.

The arithmetic is NOT here: the per-modality embed -> LayerNorm -> concat -> +PE pipeline runs
inside the HIP engine (engine.Seq2SeqEngine.embed_fwd, kernels afm_gather_rows / afm_gemm /
afm_layernorm_fwd).  These classes keep the names, constructor arguments and call shapes the
reference's callers use.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

POS_ENC_REGISTRY = {"sin_cos": "sin_cos", "learned": "learned"}  # modeling/utils.py:275 (keys only)


class CustomLMOutput(dict):
    """Seq2SeqLMOutput-like record with attribute access (modeling/utils.py:25-30)."""

    def __init__(self, loss=None, logits=None, decoder_hidden_states=None, encoder_hidden_states=None,
                 loss_dict: Optional[Dict[str, Any]] = None, **extra):
        super().__init__(loss=loss, logits=logits, decoder_hidden_states=decoder_hidden_states,
                         encoder_hidden_states=encoder_hidden_states, loss_dict=loss_dict, **extra)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class DeferredEmbedding:
    """What MultimodalEmbedding returns: the modality dict, embedded lazily INSIDE the engine so the
    embedder's backward state stays with the rest of the schedule.  `.materialize()` gives the
    (B, S, d) fp32 tensor the reference would have returned."""

    def __init__(self, owner: "MultimodalEmbedding", token_ids: Dict[str, Any]):
        self.owner, self.token_ids = owner, token_ids

    def materialize(self) -> torch.Tensor:
        eng = self.owner.engine
        x = eng.embed_fwd(self.token_ids, None)
        first = next(iter(self.token_ids.values()))
        B = (first["tokenized_input"] if isinstance(first, dict) else first).shape[0]
        return x.view(B, -1, eng.d)


class MultimodalEmbedding:
    """modeling/utils.py:44-182.  Same constructor; parameters live in the engine's flat store under
    the reference's keys `embedding.embedding_layer_dict.*`, `embedding.embedding_norm_dict.*`,
    `embedding.positional_encodings.*`."""

    def __init__(self, data_config: Dict[str, Any], d_model: int, embedding_norm: bool,
                 do_positional_encodings: bool = False, positional_encodings_type: str = "sin_cos",
                 max_seq_len: int = 1024) -> None:
        if positional_encodings_type not in POS_ENC_REGISTRY:
            raise KeyError(positional_encodings_type)
        self.data_config, self.d_model, self.embedding_norm = data_config, d_model, embedding_norm
        self.do_positional_encodings = do_positional_encodings
        self.positional_encodings_type = positional_encodings_type
        self.max_seq_len = max_seq_len
        self.engine = None  # bound by CustomModel

    def __call__(self, token_ids: Dict[str, Any]) -> DeferredEmbedding:
        if not token_ids:
            raise ValueError("At least one modality needs to be in token_ids.")
        return DeferredEmbedding(self, token_ids)

    forward = __call__
