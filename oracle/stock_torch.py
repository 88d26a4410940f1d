"""The reference's layer stack wired from STOCK torch modules.  TEST / BASELINE INFRASTRUCTURE ONLY.

`analytical_fm`'s CustomEncoderLayer / CustomDecoderLayer subclass torch.nn.TransformerEncoderLayer /
TransformerDecoderLayer (batch_first, norm_first=True, activation "gelu", eps 1e-5; reference
modeling/custom_modeling.py:108-199) and CustomEncoder / CustomDecoder stack them in nn.TransformerEncoder /
nn.TransformerDecoder with a final LayerNorm (custom_modeling.py:202-320): for the ungated configurations the
arithmetic of the reference's hot path IS these torch modules.  This file builds exactly that wiring (no
reference source involved; torch is a third-party wheel present on every box), so that

  * tests can check the op-by-op oracle (afm_oracle.py) against torch's own modules on identical weights, and
  * bench.py can time "the reference PyTorch CPU path"  next to the port.

Embedding tables / patch projections / positional encodings are taken from the oracle (they are < 0.1 % of the
FLOPs); the timed part is encoder + decoder + LM head + cross entropy + backward.
"""
from __future__ import annotations

from typing import Any, Dict

import torch
from torch import nn

from . import afm_oracle as O


class StockSeq2Seq(nn.Module):
    def __init__(self, cfg: Dict[str, Any], vocab_out: int):
        super().__init__()
        if cfg.get("gated_linear"):
            raise NotImplementedError("the gated FFN is the reference's own block, not a stock torch layer")
        d = cfg["d_model"]
        kw = dict(dropout=float(cfg.get("dropout", 0.0)), activation=cfg.get("activation_function", "gelu"), batch_first=True,
                  norm_first=bool(cfg.get("post_layer_normalisation", True)), layer_norm_eps=O.LN_EPS)    # (custom_modeling.py:122-130)
        enc = nn.TransformerEncoderLayer(d, cfg["encoder_attention_heads"], cfg["encoder_ffn_dim"], **kw)
        dec = nn.TransformerDecoderLayer(d, cfg["decoder_attention_heads"], cfg["decoder_ffn_dim"], **kw)
        self.encoder = nn.TransformerEncoder(enc, cfg["encoder_layers"], norm=nn.LayerNorm(d, eps=O.LN_EPS),
                                             enable_nested_tensor=False)
        self.decoder = nn.TransformerDecoder(dec, cfg["decoder_layers"], norm=nn.LayerNorm(d, eps=O.LN_EPS))
        self.token_ff = nn.Linear(d, vocab_out)

    def load_oracle_state(self, sd: Dict[str, torch.Tensor]) -> None:
        """The engine / oracle state dict uses the reference's key names, which ARE torch's module names."""
        own = self.state_dict()
        for k in own:
            own[k].copy_(sd[k])

    def forward(self, x_enc, attention_mask, x_dec, dec_attention_mask, labels):
        """x_enc (B,S,d) / x_dec (B,T,d): embedded inputs; masks 1 = keep; labels with -100."""
        key_pad = ~attention_mask.bool()
        mem = self.encoder(x_enc, src_key_padding_mask=key_pad)
        T = x_dec.shape[1]
        causal = torch.full((T, T), float("-inf")).triu(1)                     # custom_modeling.py:308-310
        tgt_pad = None if dec_attention_mask is None else ~dec_attention_mask.bool()
        h = self.decoder(x_dec, mem, tgt_mask=causal, tgt_key_padding_mask=tgt_pad, memory_key_padding_mask=key_pad)
        logits = self.token_ff(h)
        loss = nn.functional.cross_entropy(logits.view(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=-100)
        return logits, loss


def embed_inputs(sd, cfg, data_config, target_modality, enc, dec_ids):
    """(x_enc, x_dec) through the oracle's embedder (shared table for the decoder, positions restart at 0)."""
    x_enc = O.embed(sd, data_config, enc, cfg.get("multimodal_norm", True), cfg["positional_encoding_type"])
    x_dec = O.embed(sd, data_config, {target_modality: dec_ids}, cfg.get("multimodal_norm", True),
                    cfg["positional_encoding_type"])
    return x_enc, x_dec
