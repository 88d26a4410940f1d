"""Drop-in boundary: `HFWrapper` + `MODEL_REGISTRY` with the reference's surface
(reference modeling/wrapper.py:144-180,222-655), minus Lightning / HF / RDKit.

Same constructor keywords, same batch-dict contract (sequence-first tensors, masks True = pad,
reference data/datamodules.py:201-218), same methods the training loop calls: forward,
training_step, validation_step, predict_step, configure_optimizers, generate, _calc_token_acc.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from ..optim import FusedAdamOneCycle
from .custom_modeling import AlignConfig, CustomConfig, CustomModel
from .utils import CustomLMOutput, MultimodalEmbedding

OPTIMISER_REGISTRY = {"adam": "adam", "adamw": "adamw"}  # wrapper.py:29


def load_custom_model(model_name: str, target_tokenizer, target_modality: str, data_config: Dict[str, Any],
                      multimodal_norm: bool, **kwargs) -> Tuple[CustomModel, MultimodalEmbedding]:
    """wrapper.py:144-180."""
    engine_kw = {k: kwargs.pop(k) for k in ("device", "compute_dtype", "seed", "backward_dtype") if k in kwargs}
    known = CustomConfig.__init__.__code__.co_varnames
    cfg_kw = {k: v for k, v in kwargs.items() if k in known}
    model_config = CustomConfig.from_pretrained(
        model_name, pad_token_id=target_tokenizer.pad_token_id, bos_token_id=target_tokenizer.bos_token_id,
        eos_token_id=target_tokenizer.eos_token_id, decoder_start_token_id=target_tokenizer.bos_token_id,
        forced_eos_token_id=target_tokenizer.eos_token_id, **cfg_kw)
    emb = MultimodalEmbedding(data_config, model_config.d_model, multimodal_norm, do_positional_encodings=True,
                              positional_encodings_type=model_config.positional_encoding_type,
                              max_seq_len=model_config.max_position_embeddings)
    return CustomModel(target_modality, target_tokenizer, model_config, emb, **engine_kw), emb


MODEL_REGISTRY: Dict[str, Callable[..., Tuple[CustomModel, MultimodalEmbedding]]] = {
    "CustomModel": load_custom_model,  # the only loader on the reference's tested path
}


class SimpleTokenizerInfo:
    """The four attributes of the target tokenizer the model path reads."""

    def __init__(self, vocab_size, pad_token_id=0, bos_token_id=2, eos_token_id=3):
        self.vocab_size, self.pad_token_id = vocab_size, pad_token_id
        self.bos_token_id, self.eos_token_id = bos_token_id, eos_token_id


class HFWrapper:
    """wrapper.py:230-655."""

    def __init__(self, data_config: Dict[str, Any], model_type: str, model_name: str, target_tokenizer,
                 optimiser: str = "adam", num_steps: int = 1000, lr: float = 0.001, weight_decay: float = 0,
                 adam_beta1: float = 0.9, adam_beta2: float = 0.999, multimodal_norm: bool = True,
                 modality_dropout: Optional[List[str]] = None, **kwargs) -> None:
        if isinstance(target_tokenizer, str):
            raise NotImplementedError("pass a tokenizer object (no HF hub in this build)")
        self.target_tokenizer = target_tokenizer
        self.model_type, self.model_name, self.data_config = model_type, model_name, data_config
        self.multimodal_norm, self.modality_dropout = multimodal_norm, modality_dropout
        self.guided_generation = kwargs.get("guided_generation", False)
        self.target_modality = ""
        for modality, mc in data_config.items():
            if mc["target"]:
                self.target_modality = modality
        self.optimiser, self.lr, self.weight_decay = optimiser, lr, weight_decay
        self.adam_beta1, self.adam_beta2, self.num_steps = adam_beta1, adam_beta2, num_steps
        self.validation_step_outputs: List[Dict[str, Any]] = []
        self.clip_grad = kwargs.pop("clip_grad", 1.0)
        self.world_size = kwargs.pop("world_size", 1)
        if model_type not in MODEL_REGISTRY:
            raise KeyError(f"model_type {model_type!r}: only {list(MODEL_REGISTRY)} are built")
        self.hf_model, self.multimodal_embedding = MODEL_REGISTRY[model_type](
            model_name, target_tokenizer, self.target_modality, data_config, multimodal_norm, **kwargs)
        self.max_length = 128  # generation_config.max_length (wrapper.py:313)
        self.n_beams = kwargs.get("n_beams", 10)
        self.beam_stop_rule = "hf4"     # transformers 4.48.3 (the reference's pin); "hf5": the 5.x rule (beam.py)
        self.training = True
        self.logged: Dict[str, Any] = {}
        # _init_params (wrapper.py:320-327) happens in the engine's ParamStore.init_

    # ---- nn.Module-ish
    def train(self, mode=True):
        self.training = mode; self.hf_model.train(mode); return self

    def eval(self):
        return self.train(False)

    def state_dict(self):
        sd = {}
        for k, v in self.hf_model.state_dict().items():
            sd["hf_model." + k] = v
            if k.startswith("embedding."):
                sd["multimodal_embedding." + k[len("embedding."):]] = v  # wrapper.py:298 alias
        return sd

    def load_state_dict(self, sd, strict=True):
        inner = {k[len("hf_model."):]: v for k, v in sd.items() if k.startswith("hf_model.")}
        self.hf_model.load_state_dict(inner or sd, strict)

    def log(self, key, value, sync_dist: bool = False, **_):
        """LightningModule.log as the hot path uses it: sync_dist=True averages the scalar over the ranks
        (one tiny all-reduce per logged value, reference wrapper.py:474,486,601)."""
        if sync_dist:
            from ..trainer import sync_mean
            value = sync_mean(value)
        self.logged[key] = value

    def configure_optimizers(self):
        """wrapper.py:329-344: optimiser over all params + OneCycleLR stepped per optimiser step."""
        self.optim = FusedAdamOneCycle(self.hf_model.engine, OPTIMISER_REGISTRY[self.optimiser], lr=self.lr,
                                       weight_decay=self.weight_decay, adam_beta1=self.adam_beta1,
                                       adam_beta2=self.adam_beta2, num_steps=self.num_steps,
                                       clip_grad=self.clip_grad, world_size=self.world_size)
        return [self.optim], [{"scheduler": self.optim, "interval": "step"}]

    # ---- the hot path
    def forward(self, batch: Dict[str, Any]) -> CustomLMOutput:
        """wrapper.py:346-407."""
        input_ids = {}
        for modality, ids in batch["encoder_input"].items():
            input_ids[modality] = ({k: v.transpose(1, 0) for k, v in ids.items()} if isinstance(ids, dict)
                                   else ids.transpose(1, 0))
        decoder_input = batch["decoder_input"][self.target_modality].transpose(1, 0)
        attention_mask = (~batch["encoder_pad_mask"]).int().T
        decoder_attention_mask = (~batch["decoder_pad_mask"]).int().T
        labels = batch["target"].T.contiguous().clone()
        if isinstance(self.modality_dropout, (list, tuple)) and len(self.modality_dropout) and self.training:
            drop = np.random.choice(self.modality_dropout, np.random.randint(0, len(self.modality_dropout)),
                                    replace=False)                     # wrapper.py:368-386
            split, idx = [], 0
            for modality, ids in input_ids.items():
                n = (ids["tokenized_input"] if isinstance(ids, dict) else ids).shape[1]
                if modality not in drop:
                    split.append(attention_mask[:, idx:idx + n])
                idx += n
            for modality in drop:
                input_ids.pop(modality)
            attention_mask = torch.concat(split, dim=-1)
        labels[labels == self.target_tokenizer.pad_token_id] = -100
        inputs_embeds = self.multimodal_embedding(input_ids)
        kwargs = {}
        if "encoder_alignment_input" in batch:       # wrapper.py:395-396
            kwargs["encoder_align_target"] = batch["encoder_alignment_input"]
        return self.hf_model(inputs_embeds=inputs_embeds, attention_mask=attention_mask.contiguous(),
                             decoder_input_ids=decoder_input, decoder_attention_mask=decoder_attention_mask.contiguous(),
                             labels=labels, **kwargs)

    __call__ = forward

    def training_step(self, batch: Dict[str, Any], batch_idx: int, loss_scale: float = 1.0) -> torch.Tensor:
        """wrapper.py:455-489.  Backward runs inside (the engine is hand-scheduled); gradients
        accumulate scaled by loss_scale = 1/accumulate_grad_batches as Lightning does."""
        self.train()
        self.hf_model.backward_on_forward(True, loss_scale)
        try:
            out = self.forward(batch)
        finally:
            self.hf_model.backward_on_forward(False)
        if batch_idx % 10 == 0:
            self.log("train_loss", out.loss, sync_dist=True)
            for key, val in (out.loss_dict or {}).items():        # wrapper.py:476-487
                if val is not None:
                    self.log(f"train_{key}", val, sync_dist=True)
        return out.loss

    def _calc_token_acc(self, batch_input, model_output):
        """wrapper.py:641-655 incl. its quirk: `target` still holds pad ids, so every position counts."""
        token_ids = batch_input["target"].T
        pred = model_output.argmax if model_output.get("argmax") is not None else torch.argmax(model_output.logits, -1)
        mask = token_ids != -100
        return ((token_ids == pred) * mask).sum().float() / mask.sum().float()

    def _encode_for_generation(self, batch):
        input_ids = {m: (v.transpose(1, 0) if not isinstance(v, dict) else {k: t.transpose(1, 0) for k, t in v.items()})
                     for m, v in batch["encoder_input"].items()}
        attention_mask = (~batch["encoder_pad_mask"]).int().T.contiguous()
        enc = self.hf_model.encoder(attention_mask=attention_mask, inputs_embeds=self.multimodal_embedding(input_ids))
        return enc, attention_mask

    def generate(self, batch: Dict[str, Any], n_beams: int = 1, logits_processor=None, use_cache: bool = True,
                 graph: bool = True, device_beam: bool = True) -> torch.Tensor:
        """wrapper.py:409-453.  Encoder once, then greedy (n_beams == 1) or beam search with
        num_return_sequences = n_beams, max_length 128, forced EOS; returns (B*n_beams, <=128) ids.
        use_cache=True decodes incrementally on a device-side KV cache (engine.decode_step);
        use_cache=False reproduces the reference's full-prefix recompute (greedy only) for cross-checks."""
        if logits_processor is not None:
            raise NotImplementedError("guided generation (RDKit logits processor) is out of scope (SURVEY 2, row 18)")
        tok = self.target_tokenizer
        was = self.training
        self.eval()
        enc, attention_mask = self._encode_for_generation(batch)
        B = attention_mask.shape[0]
        dev = attention_mask.device
        eng = self.hf_model.engine
        try:
            if n_beams == 1 and not use_cache:
                ids = torch.full((B, 1), tok.bos_token_id, dtype=torch.long, device=dev)
                done = torch.zeros(B, dtype=torch.bool, device=dev)
                while ids.shape[1] < self.max_length:
                    lg = self.hf_model(encoder_outputs=enc, attention_mask=attention_mask, decoder_input_ids=ids).logits
                    ids, done = self._greedy_pick(lg[:, -1], ids, done)
                    if bool(done.all()):
                        break
                return ids
            if n_beams == 1 and graph:
                # one captured HIP graph per position (engine.decode_step_graphed); the all-finished test
                # costs a host sync, so it runs every eighth token (finished rows emit pad meanwhile and the
                # surplus all-pad columns are cut, so the ids equal the eager loop's)
                st = eng.decode_init_graphed(enc["last_hidden_state"], attention_mask, max_len=self.max_length)
                ids = torch.full((B, 1), tok.bos_token_id, dtype=torch.long, device=dev)
                done = torch.zeros(B, dtype=torch.bool, device=dev)
                while ids.shape[1] < self.max_length:
                    ids, done = self._greedy_pick(eng.decode_step_graphed(st, ids[:, -1]), ids, done)
                    if ids.shape[1] % 8 == 0 and bool(done.all()):
                        break
                keep = int((ids != tok.pad_token_id).any(0).nonzero().max()) + 1 if bool(done.all()) else ids.shape[1]
                return ids[:, :max(keep, 2)]
            st = eng.decode_init(enc["last_hidden_state"], attention_mask, beams=n_beams, max_len=self.max_length)
            if n_beams == 1:
                ids = torch.full((B, 1), tok.bos_token_id, dtype=torch.long, device=dev)
                done = torch.zeros(B, dtype=torch.bool, device=dev)
                while ids.shape[1] < self.max_length:
                    ids, done = self._greedy_pick(eng.decode_step(st, ids[:, -1]), ids, done)
                    if bool(done.all()):
                        break
                return ids
            # beam bookkeeping on the device (afm_beam_step): one small D2H every few tokens; `device_beam=False` keeps the
            # host loop (the restatement of HF's Python code the kernels are tested against)
            from ..beam import beam_search, beam_search_device
            search = beam_search_device if device_beam else beam_search
            seqs, self.last_beam_scores = search(lambda last: eng.decode_step(st, last),
                                                 lambda idx: eng.decode_reorder(st, idx), B, n_beams, eng.V,
                                                 self.max_length, tok.bos_token_id, tok.eos_token_id,
                                                 tok.pad_token_id, dev, stop_rule=self.beam_stop_rule)
            return seqs
        finally:
            self.train(was)

    def _greedy_pick(self, last_logits, ids, done):
        tok = self.target_tokenizer
        nxt = last_logits.argmax(-1)
        if ids.shape[1] == self.max_length - 1:
            nxt = torch.full_like(nxt, tok.eos_token_id)
        nxt = torch.where(done, torch.full_like(nxt, tok.pad_token_id), nxt)
        return torch.cat([ids, nxt[:, None]], 1), done | (nxt == tok.eos_token_id)

    def validation_step(self, batch: Dict[str, Any], batch_idx: int) -> Dict[str, Any]:  # noqa: ARG002
        """wrapper.py:491-525 without the RDKit Top-1 (host chemistry metric, out of scope)."""
        self.eval()
        out = self.forward(batch)
        val = {"val_loss": out.loss, "val_token_acc": self._calc_token_acc(batch, out)}
        # greedy decode of every validation batch (wrapper.py:507) scored Top-1.  The reference decodes to strings
        # and compares them verbatim (`calc_sampling_metrics(..., molecules=False)`, no canonicalisation), which for
        # an injective tokenizer is equality of the id sequences with the special tokens removed.
        gen = self.generate(batch, n_beams=1)
        top1 = self.sequence_accuracy(gen, batch["target"].T)
        val["val_molecular_accuracy_tensorboard"] = top1
        val["val_molecular_accuracy"] = top1
        for key, v in (out.loss_dict or {}).items():
            val[f"val_{key}"] = v
        self.validation_step_outputs.append(val)
        return val

    def sequence_accuracy(self, generated: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """Fraction of rows whose generated ids equal the target ids once bos / eos / pad are stripped
        (score_val_sequences, wrapper.py:604-639, with n_beams = 1 and molecules=False)."""
        tok = self.target_tokenizer
        special = (tok.pad_token_id, tok.bos_token_id, tok.eos_token_id)

        def strip(x):
            keep = torch.ones_like(x, dtype=torch.bool)
            for sp in special:
                keep &= x != sp
            keep &= x != -100
            # stable compaction: position of each kept token among the kept ones
            pos = keep.long().cumsum(1) - 1
            out = torch.full((x.shape[0], x.shape[1] + 1), -1, dtype=torch.long, device=x.device)
            out.scatter_(1, torch.where(keep, pos, torch.full_like(pos, x.shape[1])), torch.where(keep, x, torch.full_like(x, -1)))
            return out[:, :x.shape[1]], keep.sum(1)

        g, gn = strip(generated)
        t, tn = strip(target.to(generated.device))
        L = max(g.shape[1], t.shape[1])
        g = torch.nn.functional.pad(g, (0, L - g.shape[1]), value=-1)
        t = torch.nn.functional.pad(t, (0, L - t.shape[1]), value=-1)
        return ((g == t).all(1) & (gn == tn)).float().mean().reshape(1)

    def on_validation_epoch_end(self):
        """wrapper.py:527-530,580-603: average every metric over the validation batches, log with sync_dist."""
        if not self.validation_step_outputs:
            return {}
        keys = list(self.validation_step_outputs[0])
        avg = {}
        for k in keys:
            vals = [o[k] for o in self.validation_step_outputs]
            if any(v is None for v in vals):
                continue
            avg[k] = sum(vals) / len(vals)
        for k, v in avg.items():
            self.log(k, v, sync_dist=True)
        self.validation_step_outputs = []
        # what ModelCheckpoint / EarlyStopping monitor in the reference is the LOGGED (rank-averaged) value, the same on every rank
        return {k: self.logged[k] for k in avg}

    def predict_step(self, batch, batch_idx):  # noqa: ARG002
        """wrapper.py:532-578 (greedy ids instead of decoded beam strings)."""
        self.eval()
        out = self.forward(batch)
        return {"loss": out.loss, "predictions": self.generate(batch, n_beams=self.n_beams), "targets": batch.get("target_smiles")}
