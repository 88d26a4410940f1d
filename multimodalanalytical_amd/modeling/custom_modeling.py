"""Host-side mirror of `analytical_fm.modeling.custom_modeling` (reference
modeling/custom_modeling.py): CustomConfig, AlignConfig, CustomModel with the same constructor and
`forward` keywords, backed by the HIP engine instead of torch.nn.Transformer*."""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from ..engine import Seq2SeqEngine
from .utils import CustomLMOutput, DeferredEmbedding, MultimodalEmbedding


class AlignConfig:
    """custom_modeling.py:18-37."""

    def __init__(self, align_network, hidden_dimension, conv_channels, kernel_size, output_dimension,
                 loss_lambda, loss_function):
        self.align_network, self.hidden_dimension = align_network, hidden_dimension
        self.conv_channels, self.kernel_size = conv_channels, kernel_size
        self.output_dimension, self.loss_lambda, self.loss_function = output_dimension, loss_lambda, loss_function


class CustomConfig:
    """custom_modeling.py:40-105, same fields and defaults.  `from_pretrained(model_name, **kw)` keeps
    the reference call shape (wrapper.py:153-162) but never touches the HF hub: the only inherited
    bart-base fields that matter (dropout 0.1, gelu, is_encoder_decoder) equal these defaults."""

    def __init__(self, d_model=512, max_position_embeddings=1024, encoder_layers=6, encoder_attention_heads=8,
                 encoder_ffn_dim=2048, decoder_layers=6, decoder_attention_heads=8, decoder_ffn_dim=2048,
                 dropout=0.1, activation_function="gelu", post_layer_normalisation=True, gated_linear=False,
                 positional_encoding_type="sin_cos", bos_token_id=2, eos_token_id=3, pad_token_id=0,
                 decoder_start_token_id=2, forced_eos_token_id=3, guided_generation=False, align_config=None,
                 **kwargs):
        self.d_model, self.max_position_embeddings = d_model, max_position_embeddings
        self.encoder_layers, self.encoder_attention_heads, self.encoder_ffn_dim = encoder_layers, encoder_attention_heads, encoder_ffn_dim
        self.decoder_layers, self.decoder_attention_heads, self.decoder_ffn_dim = decoder_layers, decoder_attention_heads, decoder_ffn_dim
        self.dropout, self.activation_function = dropout, activation_function
        self.gated_linear, self.post_layer_normalisation = gated_linear, post_layer_normalisation
        self.positional_encoding_type = positional_encoding_type
        self.bos_token_id, self.eos_token_id, self.pad_token_id = bos_token_id, eos_token_id, pad_token_id
        self.decoder_start_token_id, self.forced_eos_token_id = decoder_start_token_id, forced_eos_token_id
        self.guided_generation = guided_generation
        if align_config and not isinstance(align_config, AlignConfig):
            align_config = AlignConfig(**align_config)
        self.align_config = align_config
        self.is_encoder_decoder = True
        self.extra = kwargs
        if activation_function not in ("gelu", "relu"):     # torch's string options for the layers (custom_modeling.py:127,174)
            raise NotImplementedError(f"activation_function {activation_function!r}: 'gelu' (reference default) and 'relu' are built")

    @classmethod
    def from_pretrained(cls, model_name: str, **kwargs):  # noqa: ARG003 - name kept, no hub lookup
        return cls(**kwargs)

    def to_dict(self) -> Dict[str, Any]:
        keys = ("d_model max_position_embeddings encoder_layers encoder_attention_heads encoder_ffn_dim "
                "decoder_layers decoder_attention_heads decoder_ffn_dim dropout gated_linear "
                "positional_encoding_type post_layer_normalisation activation_function").split()
        out = {k: getattr(self, k) for k in keys}
        if self.align_config is not None:
            out["align_config"] = dict(vars(self.align_config))
        return out


class _EncoderHandle:
    """`hf_model.encoder(attention_mask=, inputs_embeds=)` (wrapper.py:433-441)."""

    def __init__(self, model):
        self.model = model

    def __call__(self, inputs_embeds=None, attention_mask=None):
        eng = self.model.engine
        enc = inputs_embeds.token_ids if isinstance(inputs_embeds, DeferredEmbedding) else inputs_embeds    # dict or (B, S, d) tensor
        mem, _ = eng.encode(enc, attention_mask)
        B, S = attention_mask.shape
        from ..x2 import X2
        hs = mem if isinstance(mem, X2) else mem.view(B, S, eng.d)   # split-pair memory stays the engine's 2-D object
        return {"last_hidden_state": hs, "attention_mask": attention_mask}


class CustomModel:
    """custom_modeling.py:323-508."""

    def __init__(self, target_modality, target_tokenizer, config: CustomConfig,
                 multimodal_embedding_layer: MultimodalEmbedding, device="cuda:0",
                 compute_dtype=torch.bfloat16, seed: int = 3247, backward_dtype=None):
        self.config = config
        self.target_modality = target_modality
        self.decoder_vocab_size = target_tokenizer.vocab_size
        self.embedding = multimodal_embedding_layer
        cfg = config.to_dict()
        cfg["multimodal_norm"] = multimodal_embedding_layer.embedding_norm
        self.engine = Seq2SeqEngine(cfg, multimodal_embedding_layer.data_config, target_modality,
                                    self.decoder_vocab_size, device=device, compute_dtype=compute_dtype, seed=seed,
                                    backward_dtype=backward_dtype)
        multimodal_embedding_layer.engine = self.engine
        self.encoder = _EncoderHandle(self)
        self._grad_enabled, self._loss_scale = False, 1.0

    # -- torch.nn.Module-like conveniences
    def train(self, mode=True):
        self.engine.train(mode); return self

    def eval(self):
        return self.train(False)

    def state_dict(self):
        return self.engine.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.engine.load_state_dict(sd, strict)

    def backward_on_forward(self, enabled: bool, loss_scale: float = 1.0):
        """The engine runs backward inside forward (hand-scheduled, no autograd graph): the training
        loop arms it for the next call; grads ACCUMULATE in the flat buffer scaled by loss_scale."""
        self._grad_enabled, self._loss_scale = enabled, loss_scale

    def forward(self, inputs_embeds=None, attention_mask=None, encoder_outputs: Optional[Dict[str, Any]] = None,
                decoder_input_ids=None, decoder_attention_mask=None, labels=None, use_cache=False,
                return_dict=False, encoder_align_target=None) -> CustomLMOutput:
        eng = self.engine
        generating = isinstance(encoder_outputs, dict)
        memory, enc_inputs = None, None
        if generating:
            B_, S_ = attention_mask.shape
            memory = eng._mem_rows(encoder_outputs["last_hidden_state"], B_, S_)
        else:
            # this model's MultimodalEmbedding output (embedded inside the engine, trainable) or, as in the reference
            # (custom_modeling.py:420-445), any (B, S, d) tensor of embeddings: forward / generate only
            if isinstance(inputs_embeds, DeferredEmbedding):
                enc_inputs = inputs_embeds.token_ids
            elif torch.is_tensor(inputs_embeds):
                enc_inputs = inputs_embeds
            else:
                raise TypeError("inputs_embeds: MultimodalEmbedding output or a (B, S, d) tensor")
        out = eng.forward(enc_inputs, attention_mask, decoder_input_ids, decoder_attention_mask, labels,
                          backward=self._grad_enabled and labels is not None and not generating,
                          loss_scale=self._loss_scale, memory=memory,
                          encoder_align_target=None if generating else encoder_align_target)
        loss = out.get("loss")
        loss_dict = None if loss is None else out.get("loss_dict", {"model_only_loss": loss, "alignment_loss": None})
        return CustomLMOutput(loss=loss, logits=out["logits"], decoder_hidden_states=None,
                              encoder_hidden_states=out["encoder_hidden_states"], loss_dict=loss_dict,
                              argmax=out.get("argmax"), encoder_row_map=out.get("encoder_row_map"))

    __call__ = forward
