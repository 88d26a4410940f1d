"""CPU oracle for the spectra->SMILES training path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, op by op on CPU tensors, of the arithmetic
the reference (rxn4chemistry/MultimodalAnalytical, `analytical_fm`) runs on its hot
path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import it; the product (`multimodalanalytical_amd`) never does.

Parity status: the reference holds no numeric golden for this path (its tests check
exit codes only), so this oracle is pinned against outputs of the
reference itself, generated in the build container by `oracle/make_goldens.py`
(imports `/root/reference/src`) and committed under `tests/golden/`.
`tests/test_oracle_golden.py` checks every function here against those vectors.

What is restated (reference file:line):
  * MultimodalEmbedding.forward            modeling/utils.py:142-182
  * SincCosPositionalEncoding              modeling/utils.py:198-239
  * LearnedPositionalEncoding              modeling/utils.py:242-272
  * CustomEncoder / CustomEncoderLayer     modeling/custom_modeling.py:108-152,202-243
  * CustomDecoder / CustomDecoderLayer     modeling/custom_modeling.py:155-199,246-320
  * CustomModel.forward (LM head + CE)     modeling/custom_modeling.py:420-508
  * HFWrapper.forward batch re-layout      modeling/wrapper.py:346-407
  * HFWrapper._calc_token_acc              modeling/wrapper.py:641-655
  * HFWrapper.configure_optimizers         modeling/wrapper.py:329-344  (AdamW/Adam + OneCycleLR)
  * Lightning clip_grad_norm / accumulate  trainer/trainer.py:60-72
The layer equations that live in third-party torch (nn.TransformerEncoderLayer,
nn.TransformerDecoderLayer, nn.MultiheadAttention, F.scaled_dot_product_attention,
nn.LayerNorm, F.gelu, nn.CrossEntropyLoss, optim.AdamW, OneCycleLR; torch 2.7.1 pinned
by the reference's uv.lock) are written out here from their documented definitions:
matmul, exp, erf, mean/var; nothing here calls torch.nn layers or torch.optim.

The backward pass is torch autograd over these explicit forward ops (the reference's
backward is autograd too).  All maths is in the dtype of the parameters handed in
(float32 for parity with the reference, float64 for a tighter yardstick).
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Tuple

import torch

TEXT_TYPES = (
    "text", "text_spectrum", "peak_positional_encoding", "run_length_encoding",
    "multiplets", "carbon", "msms_text",
)  # modeling/utils.py:93-101
PATCH_TYPES = ("1D_patches", "msms_number")  # modeling/utils.py:107
LN_EPS = 1e-5  # nn.LayerNorm default, used everywhere in the reference


# ----------------------------------------------------------------------------- primitives
def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = LN_EPS) -> torch.Tensor:
    """nn.LayerNorm over the last dim: biased variance, eps inside the sqrt."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * w + b


def gelu(x: torch.Tensor) -> torch.Tensor:
    """Exact (erf) GELU: `activation_function: "gelu"` -> F.gelu (custom_modeling.py:54)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


def sincos_table(d_model: int, max_len: int, dtype=torch.float32) -> torch.Tensor:
    """modeling/utils.py:226-239: pe[p, 2i] = sin(p / 10000^(2i/d)), pe[p, 2i+1] = cos(.)."""
    frac = (torch.arange(0, d_model, 2, dtype=torch.float64) / d_model).float()
    div = 10000 ** frac                                 # float32 pow, like the reference
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    ang = pos * div.reciprocal()          # `int / tensor` in the reference is reciprocal-then-multiply
    pe = torch.stack((torch.sin(ang), torch.cos(ang)), dim=2).flatten(1)[:, :d_model]
    return pe.to(dtype)


def attention(
    q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
    key_pad: Optional[torch.Tensor], causal: bool,
) -> torch.Tensor:
    """softmax(Q K^T / sqrt(dh) + mask) V per head; q (B,h,Tq,dh), k/v (B,h,Tk,dh).

    key_pad (B,Tk) bool True = padded key (-> -inf); causal adds triu(-inf, 1)
    (custom_modeling.py:308-310).  Rows with every key masked give zeros
    (torch `_safe_softmax`).
    """
    dh = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    neg = torch.finfo(s.dtype).min
    masked = torch.zeros(s.shape, dtype=torch.bool)
    if key_pad is not None:
        masked = masked | key_pad[:, None, None, :]
    if causal:
        tq, tk = s.shape[-2], s.shape[-1]
        masked = masked | torch.ones(tq, tk, dtype=torch.bool).triu(1)
    s = s.masked_fill(masked, neg)
    m = s.max(dim=-1, keepdim=True).values
    e = torch.exp(s - m).masked_fill(masked, 0.0)
    den = e.sum(dim=-1, keepdim=True)
    p = torch.where(den > 0, e / den.clamp_min(1e-38), torch.zeros_like(e))
    return p @ v


def mha(
    x_q: torch.Tensor, x_kv: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, n_heads: int,
    key_pad: Optional[torch.Tensor], causal: bool,
) -> torch.Tensor:
    """nn.MultiheadAttention(batch_first=True): packed in-proj, heads, SDPA, out-proj."""
    w_in, b_in = sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"]
    d = x_q.shape[-1]
    q = linear(x_q, w_in[:d], b_in[:d])
    k = linear(x_kv, w_in[d:2 * d], b_in[d:2 * d])
    v = linear(x_kv, w_in[2 * d:], b_in[2 * d:])
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    dh = d // n_heads
    q = q.view(B, Tq, n_heads, dh).transpose(1, 2)
    k = k.view(B, Tk, n_heads, dh).transpose(1, 2)
    v = v.view(B, Tk, n_heads, dh).transpose(1, 2)
    o = attention(q, k, v, key_pad, causal).transpose(1, 2).reshape(B, Tq, d)
    return linear(o, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def ffn(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, gated: bool, act: str = "gelu") -> torch.Tensor:
    """W2 act(W1 x) or W2 (act(W1 x) * (Wg x)) (custom_modeling.py:137-152,184-199); act = config.activation_function, which the
    layers hand to torch (custom_modeling.py:127,174): "gelu" (exact erf form) or "relu"."""
    u = linear(x, sd[prefix + "linear1.weight"], sd[prefix + "linear1.bias"])
    if act not in ("gelu", "relu"):
        raise ValueError(f"activation {act!r}")
    h = gelu(u) if act == "gelu" else torch.clamp(u, min=0)
    if gated:
        h = h * linear(x, sd[prefix + "gate.weight"], sd[prefix + "gate.bias"])
    return linear(h, sd[prefix + "linear2.weight"], sd[prefix + "linear2.bias"])


# ----------------------------------------------------------------------------- embedding
def _embed_one(sd, prefix: str, mcfg: Dict[str, Any], x) -> torch.Tensor:
    """One modality through its embedder (modeling/utils.py:84-140,152-162)."""
    lp = f"{prefix}embedding_layer_dict."
    name = mcfg["_name"]
    scale = None
    if isinstance(x, dict):  # xVal: token embedding times numerical value (utils.py:154-160)
        scale = x["numerical_values"]
        x = x["tokenized_input"]
    if mcfg["type"] in TEXT_TYPES:
        e = sd[f"{lp}{name}.weight"][x]
    elif mcfg["type"] in PATCH_TYPES:
        enc = mcfg.get("preprocessor_arguments", {}).get("encoding_type", "linear")
        x = x.to(sd[f"{lp}{name}.weight" if enc == "linear" else f"{lp}{name}.0.weight"].dtype)
        if enc == "linear":
            e = linear(x, sd[f"{lp}{name}.weight"], sd[f"{lp}{name}.bias"])
        elif enc == "linear_2_layer":
            e = torch.relu(linear(x, sd[f"{lp}{name}.0.weight"], sd[f"{lp}{name}.0.bias"]))
            e = linear(e, sd[f"{lp}{name}.2.weight"], sd[f"{lp}{name}.2.bias"])
        elif enc == "linear_3_layer":
            e = torch.relu(linear(x, sd[f"{lp}{name}.0.weight"], sd[f"{lp}{name}.0.bias"]))
            e = torch.relu(linear(e, sd[f"{lp}{name}.2.weight"], sd[f"{lp}{name}.2.bias"]))
            e = linear(e, sd[f"{lp}{name}.4.weight"], sd[f"{lp}{name}.4.bias"])
        else:
            raise NotImplementedError(enc)
    else:
        raise NotImplementedError(mcfg["type"])
    if scale is not None:
        e = e * scale.unsqueeze(-1).to(e.dtype)
    return e


def embed(
    sd: Dict[str, torch.Tensor], data_config: Dict[str, Any], inputs: Dict[str, Any],
    embedding_norm: bool = True, pos_type: str = "sin_cos", prefix: str = "embedding.",
) -> torch.Tensor:
    """MultimodalEmbedding.forward (modeling/utils.py:142-182): per-modality embed ->
    per-modality LayerNorm -> concat on the sequence dim -> + positional encodings
    (positions run over the concatenated sequence, starting at 0)."""
    parts: List[torch.Tensor] = []
    for name, x in inputs.items():
        mcfg = dict(data_config[name]); mcfg["_name"] = name
        e = _embed_one(sd, prefix, mcfg, x)
        if embedding_norm:
            e = layer_norm(e, sd[f"{prefix}embedding_norm_dict.{name}.weight"],
                           sd[f"{prefix}embedding_norm_dict.{name}.bias"])
        parts.append(e)
    x = torch.cat(parts, dim=1)
    S = x.shape[1]
    if pos_type == "sin_cos":
        pe = sd[f"{prefix}positional_encodings.pos_enc"][:S]
    elif pos_type == "learned":  # LayerNorm(Embedding[arange(S)]) (utils.py:257-272)
        pe = layer_norm(sd[f"{prefix}positional_encodings.pos_encodings.weight"][:S],
                        sd[f"{prefix}positional_encodings.norm.weight"],
                        sd[f"{prefix}positional_encodings.norm.bias"])
    else:
        raise KeyError(pos_type)
    return x + pe.unsqueeze(0).to(x.dtype)


# ----------------------------------------------------------------------------- model
def encoder(sd, cfg: Dict[str, Any], x: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """CustomEncoder.forward (custom_modeling.py:220-243): N pre-LN layers + final LN.
    attention_mask (B,S) 1 = keep; gate is always applied (training semantics, A.1)."""
    key_pad = ~attention_mask.bool()
    act = cfg.get("activation_function", "gelu")
    pre = cfg.get("post_layer_normalisation", True)      # the reference's flag IS torch's norm_first (custom_modeling.py:129, A.2)
    for i in range(cfg["encoder_layers"]):
        p = f"encoder.layers.{i}."
        if pre:
            h = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
            x = x + mha(h, h, sd, p + "self_attn.", cfg["encoder_attention_heads"], key_pad, False)
            h = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
            x = x + ffn(h, sd, p, cfg["gated_linear"], act)
        else:   # torch:nn/modules/transformer.py norm_first=False: x = norm1(x + sa(x)); x = norm2(x + ff(x))
            x = layer_norm(x + mha(x, x, sd, p + "self_attn.", cfg["encoder_attention_heads"], key_pad, False),
                           sd[p + "norm1.weight"], sd[p + "norm1.bias"])
            x = layer_norm(x + ffn(x, sd, p, cfg["gated_linear"], act), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    return layer_norm(x, sd["encoder.norm.weight"], sd["encoder.norm.bias"])


def decoder(sd, cfg, data_config, target_modality: str, dec_ids: torch.Tensor,
            memory: torch.Tensor, enc_attention_mask: torch.Tensor,
            dec_attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """CustomDecoder.forward (custom_modeling.py:271-320): shared embedding (positions
    restart at 0), causal + key-pad self-attn, cross-attn with memory key-pad, FFN."""
    x = embed(sd, data_config, {target_modality: dec_ids}, cfg.get("multimodal_norm", True),
              cfg["positional_encoding_type"])
    tgt_pad = None if dec_attention_mask is None else ~dec_attention_mask.bool()
    mem_pad = ~enc_attention_mask.bool()
    act = cfg.get("activation_function", "gelu")
    pre = cfg.get("post_layer_normalisation", True)
    H = cfg["decoder_attention_heads"]
    for i in range(cfg["decoder_layers"]):
        p = f"decoder.layers.{i}."
        if pre:
            h = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
            x = x + mha(h, h, sd, p + "self_attn.", H, tgt_pad, True)
            h = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
            x = x + mha(h, memory, sd, p + "multihead_attn.", H, mem_pad, False)
            h = layer_norm(x, sd[p + "norm3.weight"], sd[p + "norm3.bias"])
            x = x + ffn(h, sd, p, cfg["gated_linear"], act)
        else:
            x = layer_norm(x + mha(x, x, sd, p + "self_attn.", H, tgt_pad, True), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
            x = layer_norm(x + mha(x, memory, sd, p + "multihead_attn.", H, mem_pad, False), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
            x = layer_norm(x + ffn(x, sd, p, cfg["gated_linear"], act), sd[p + "norm3.weight"], sd[p + "norm3.bias"])
    return layer_norm(x, sd["decoder.norm.weight"], sd["decoder.norm.bias"])


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """nn.CrossEntropyLoss(): mean over labels != -100 of -log_softmax[label]."""
    V = logits.shape[-1]
    lg = logits.reshape(-1, V)
    lb = labels.reshape(-1)
    m = lg.max(dim=-1, keepdim=True).values
    lse = m.squeeze(-1) + torch.log(torch.exp(lg - m).sum(dim=-1))
    keep = lb != -100
    picked = lg.gather(1, lb.clamp_min(0).unsqueeze(1)).squeeze(1)
    return ((lse - picked) * keep).sum() / keep.sum()


def align_head(sd: Dict[str, torch.Tensor], acfg: Dict[str, Any], memory: torch.Tensor,
               attention_mask: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Encoder alignment loss (custom_modeling.py:363-396 network, 453-475 use): masked mean of the
    encoder output over kept tokens -> MLP (or Linear-ReLU-Linear-Conv1d-ReLU-Conv1d on a length-1
    sequence: with padding k//2 only the centre tap of the first convolution sees data) -> sigmoid ->
    mse / mae / sid against the (B, output_dimension) target spectrum."""
    mask = attention_mask.to(memory.dtype).unsqueeze(-1)
    pooled = (memory * mask).sum(dim=1) / mask.sum(dim=1)
    h = torch.relu(linear(pooled, sd["align_network.0.weight"], sd["align_network.0.bias"]))
    if acfg["align_network"] == "mlp":
        z = linear(h, sd["align_network.2.weight"], sd["align_network.2.bias"])
    else:
        h = linear(h, sd["align_network.2.weight"], sd["align_network.2.bias"])
        k = acfg["kernel_size"]
        h = torch.relu(linear(h, sd["align_network.4.weight"][:, :, k // 2], sd["align_network.4.bias"]))
        z = linear(h, sd["align_network.6.weight"][:, :, 0], sd["align_network.6.bias"])
    pred = torch.sigmoid(z)
    fn = acfg["loss_function"]
    if fn == "mse":
        return ((pred - target) ** 2).mean()
    if fn == "mae":
        return (pred - target).abs().mean()
    if fn == "sid":   # the reference's OWN kl_div (modeling/utils.py:8-22), not F.kl_div: both arguments are
        # clamped to >= 1e-16, kl = p * log(p / q), "batchmean" = sum / B; sid = kl(pred, t) + kl(t, pred)
        B = pred.shape[0]
        p, q = pred.clamp(min=1e-16), target.clamp(min=1e-16)
        return (p * (p / q).log()).sum() / B + (q * (q / p).log()).sum() / B
    raise ValueError(f"Loss function {fn} not supported for alignment")


def model_forward(
    sd: Dict[str, torch.Tensor], cfg: Dict[str, Any], data_config: Dict[str, Any],
    target_modality: str, enc_inputs: Dict[str, Any], attention_mask: torch.Tensor,
    dec_ids: torch.Tensor, dec_attention_mask: Optional[torch.Tensor],
    labels: Optional[torch.Tensor] = None, memory: Optional[torch.Tensor] = None,
    encoder_align_target: Optional[torch.Tensor] = None,
) -> Dict[str, Any]:
    """HFWrapper.forward's model call (wrapper.py:392-405) -> CustomModel.forward
    (custom_modeling.py:420-508).  Batch-first inputs; labels already hold -100 on pads."""
    if memory is None:
        x = embed(sd, data_config, enc_inputs, cfg.get("multimodal_norm", True),
                  cfg["positional_encoding_type"])
        memory = encoder(sd, cfg, x, attention_mask)
    dec = decoder(sd, cfg, data_config, target_modality, dec_ids, memory, attention_mask,
                  dec_attention_mask)
    logits = linear(dec, sd["token_ff.weight"], sd["token_ff.bias"])
    out = {"logits": logits, "encoder_hidden_states": memory, "decoder_hidden_states": dec}
    if labels is not None:
        out["loss"] = cross_entropy(logits, labels)
        if cfg.get("align_config") and encoder_align_target is not None:   # total = lm + lambda * align
            al = align_head(sd, cfg["align_config"], memory, attention_mask, encoder_align_target)
            out["loss_dict"] = {"model_only_loss": out["loss"], "alignment_loss": al}
            out["loss"] = out["loss"] + cfg["align_config"]["loss_lambda"] * al
    return out


def batch_to_model_inputs(batch: Dict[str, Any], target_modality: str, pad_token_id: int = 0):
    """HFWrapper.forward's re-layout (wrapper.py:356-365,389): seq-first -> batch-first,
    pad masks (True = pad) -> attention masks (1 = keep), labels pad -> -100."""
    enc = {}
    for m, v in batch["encoder_input"].items():
        if isinstance(v, dict):
            enc[m] = {k: t.transpose(1, 0) for k, t in v.items()}
        else:
            enc[m] = v.transpose(1, 0)
    dec_ids = batch["decoder_input"][target_modality].transpose(1, 0)
    attention_mask = (~batch["encoder_pad_mask"]).int().T
    dec_mask = (~batch["decoder_pad_mask"]).int().T
    labels = batch["target"].T.contiguous().clone()
    labels[labels == pad_token_id] = -100
    return enc, attention_mask, dec_ids, dec_mask, labels


def token_accuracy(logits: torch.Tensor, target_batch_first: torch.Tensor) -> torch.Tensor:
    """HFWrapper._calc_token_acc (wrapper.py:641-655).  `target` still holds pad ids (the
    -100 substitution happened on a copy), so its `!= -100` mask is all True (A.4)."""
    pred = torch.argmax(logits, dim=-1)
    mask = target_batch_first != -100
    return ((pred == target_batch_first) * mask).sum().float() / mask.sum().float()


# ----------------------------------------------------------------------------- optimiser
def onecycle(step: int, total_steps: int, max_lr: float, pct_start: float = 0.3,
             div_factor: float = 25.0, final_div_factor: float = 1e4,
             base_momentum: float = 0.85, max_momentum: float = 0.95) -> Tuple[float, float]:
    """torch OneCycleLR(optim, max_lr, total_steps) with defaults (wrapper.py:341): cosine
    anneal, two phases, cycle_momentum=True => beta1 is cycled 0.95 -> 0.85 -> 0.95 and the
    configured adam_beta1 is ignored.  Returns (lr, beta1) in force for
    optimiser step number `step` (0-based)."""
    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)
    initial_lr = max_lr / div_factor
    min_lr = initial_lr / final_div_factor
    end1 = float(pct_start * total_steps) - 1.0
    end2 = float(total_steps) - 1.0
    if step <= end1:
        pct = step / end1 if end1 != 0 else 0.0
        return cos(initial_lr, max_lr, pct), cos(max_momentum, base_momentum, pct)
    pct = (step - end1) / (end2 - end1)
    return cos(max_lr, min_lr, pct), cos(base_momentum, max_momentum, pct)


def clip_grad_norm(grads: List[torch.Tensor], max_norm: float) -> Tuple[torch.Tensor, float]:
    """torch.nn.utils.clip_grad_norm_ (Lightning gradient_clip_val, trainer.py:65):
    total L2 norm; scale = min(1, max_norm / (norm + 1e-6)); in place."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).to(grads[0].dtype)
    coef = min(1.0, float(max_norm / (total + 1e-6)))
    for g in grads:
        g.mul_(coef)
    return total, coef


def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, t: int,
              lr: float, beta1: float, beta2: float, eps: float, weight_decay: float,
              decoupled: bool) -> None:
    """One torch.optim.Adam (decoupled=False: L2 into the gradient) or AdamW
    (decoupled=True: p *= 1 - lr*wd) update, t = 1-based step count; in place."""
    if weight_decay != 0.0:
        if decoupled:
            p.mul_(1.0 - lr * weight_decay)
        else:
            g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


class OracleTrainer:
    """Accumulate-k micro-batches, clip, Adam(W) + OneCycle: the Lightning automatic
    optimisation loop the reference configures (trainer/trainer.py:60-72,
    wrapper.py:329-344), over a state dict of leaf tensors."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg, data_config, target_modality: str,
                 lr: float, total_steps: int, optimiser: str = "adamw", weight_decay: float = 0.0,
                 beta2: float = 0.999, eps: float = 1e-8, acc_batches: int = 4, clip: float = 1.0,
                 buffers: Tuple[str, ...] = ("embedding.positional_encodings.pos_enc",)):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.cfg, self.dc, self.tm = cfg, data_config, target_modality
        self.names = [k for k in self.sd if k not in buffers]
        for k in self.names:
            self.sd[k].requires_grad_(True)
        self.m = {k: torch.zeros_like(self.sd[k]) for k in self.names}
        self.v = {k: torch.zeros_like(self.sd[k]) for k in self.names}
        self.lr, self.total = lr, total_steps
        self.decoupled = optimiser == "adamw"
        self.wd, self.beta2, self.eps = weight_decay, beta2, eps
        self.acc, self.clip = acc_batches, clip
        self.step_count = 0
        self.micro = 0

    def micro_batch(self, enc, am, dec_ids, dm, labels, encoder_align_target=None) -> Dict[str, Any]:
        out = model_forward(self.sd, self.cfg, self.dc, self.tm, enc, am, dec_ids, dm, labels,
                            encoder_align_target=encoder_align_target)
        (out["loss"] / self.acc).backward()
        self.micro += 1
        if self.micro % self.acc == 0:
            self.optimizer_step()
        return out

    def optimizer_step(self) -> None:
        with torch.no_grad():
            names = [k for k in self.names if self.sd[k].grad is not None]
            grads = [self.sd[k].grad for k in names]
            self.last_norm, _ = clip_grad_norm(grads, self.clip)
            lr, beta1 = onecycle(self.step_count, self.total, self.lr)
            self.step_count += 1
            for k in names:
                adam_step(self.sd[k], self.sd[k].grad, self.m[k], self.v[k], self.step_count,
                          lr, beta1, self.beta2, self.eps, self.wd, self.decoupled)
            for k in self.names:
                self.sd[k].grad = None


# ----------------------------------------------------------------------------- decode
def greedy_decode(sd, cfg, data_config, target_modality, enc_inputs, attention_mask,
                  max_length: int = 128, bos: int = 2, eos: int = 3, pad: int = 0) -> torch.Tensor:
    """Greedy decode with full-prefix recompute, the loop HF generate(num_beams=1,
    use_cache=False, forced_eos) runs over CustomModel.forward(encoder_outputs=...)
    (wrapper.py:409-453, custom_modeling.py:447-483): argmax of the last position,
    finished rows emit pad, token max_length-1 is forced to eos."""
    with torch.no_grad():
        x = embed(sd, data_config, enc_inputs, cfg.get("multimodal_norm", True),
                  cfg["positional_encoding_type"])
        memory = encoder(sd, cfg, x, attention_mask)
        B = memory.shape[0]
        ids = torch.full((B, 1), bos, dtype=torch.long)
        done = torch.zeros(B, dtype=torch.bool)
        while ids.shape[1] < max_length:
            out = model_forward(sd, cfg, data_config, target_modality, None, attention_mask,
                                ids, None, None, memory=memory)
            nxt = out["logits"][:, -1].argmax(-1)
            if ids.shape[1] == max_length - 1:
                nxt = torch.full_like(nxt, eos)
            nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
            ids = torch.cat([ids, nxt[:, None]], dim=1)
            done = done | (nxt == eos)
            if bool(done.all()):
                break
        return ids


# ---------------------------------------------------------------- input path
def patch_preprocess(spectra, present, mean: float, std: float, patch_size: int, masking: bool = False,
                     interpolation: bool = False, overlap: int = 1, derivative: bool = False):
    """CPU restatement of PatchPreprocessor.__call__ (data/preprocessing/patches.py:54-107), numpy, written
    out element by element in the order the reference computes.

    spectra (B, L) float32, present (B,) bool (False = the reference's `None` row: zeros before
    standardisation, patches.py:63-67).  Returns (patches (B, P, ps) float32, mask (B, P) bool, True = pad).
      * interpolation (patches.py:47-52): scipy interp1d (linear) from the grid 400, 402, .. (L points) to
        650, 652, .. 3898: x_new = x_old[i + 125] exactly, interval (lo, hi) = (i + 124, i + 125), value
        slope * (x_new - x_lo) + y_lo in float64 with slope = (y_hi - y_lo) / 2; the first point has
        lo = 0.  torch.Tensor(...) then rounds to float32 (patches.py:73).
      * standardise in float32: (x - float32(mean)) / float32(std) (patches.py:76).
      * trim to whole patches, view (P, ps) or unfold with step ps // overlap (patches.py:79-90).
      * derivative (patches.py:92-96): torch.gradient of the RAW float32 spectrum (central differences,
        one-sided at the ends), trimmed and patched, appended after the spectrum's patches.
      * mask (patches.py:99-105): masking -> patch sum == 0, else all-True rows for absent spectra.
    """
    import numpy as np
    spectra = np.asarray(spectra, dtype=np.float32)
    present = np.asarray(present, dtype=bool)
    B, L = spectra.shape
    raw = np.where(present[:, None], spectra, np.float32(0.0)).astype(np.float32)
    if interpolation:
        y = raw.astype(np.float64)
        n_new = (3900 - 650 + 1) // 2                     # np.arange(650, 3900, 2): 1625 points
        out = np.empty((B, n_new), dtype=np.float64)
        for i in range(n_new):
            hi = max(i + 125, 1)
            lo = hi - 1
            x_new = 650.0 + 2.0 * i
            x_lo = 400.0 + 2.0 * lo
            slope = (y[:, hi] - y[:, lo]) / 2.0
            out[:, i] = slope * (x_new - x_lo) + y[:, lo]
        raw = out.astype(np.float32)
    stdz = ((raw - np.float32(mean)) / np.float32(std)).astype(np.float32)
    n_patches = stdz.shape[1] // patch_size
    trim = n_patches * patch_size
    if overlap == 1:
        patched = stdz[:, :trim].reshape(B, n_patches, patch_size)
    else:
        step = patch_size // overlap
        n_unf = (trim - patch_size) // step + 1
        patched = np.stack([stdz[:, k * step:k * step + patch_size] for k in range(n_unf)], axis=1)
    if derivative:
        g = np.empty_like(raw)
        g[:, 1:-1] = (raw[:, 2:] - raw[:, :-2]) / np.float32(2.0)
        g[:, 0] = raw[:, 1] - raw[:, 0]
        g[:, -1] = raw[:, -1] - raw[:, -2]
        patched = np.concatenate([patched, g[:, :trim].reshape(B, n_patches, patch_size)], axis=1)
    if masking:
        mask = patched.sum(-1, dtype=np.float32) == 0
    else:
        mask = np.repeat(~present[:, None], patched.shape[1], axis=1)
    return patched.astype(np.float32), mask


def normalize_spectrum(spectrum):
    """data/datasets.py:49-56, list arithmetic in Python floats: min / max are taken BEFORE the negatives
    are clipped, then (x - min) / (max - min); a flat spectrum becomes zeros."""
    mn, mx = min(spectrum), max(spectrum)
    spectrum = [max(0, x) for x in spectrum]
    if mx - mn == 0:
        return [0] * len(spectrum)
    return [(x - mn) / (mx - mn) for x in spectrum]


def mix_indices(n_rows: int, mix_config: Dict[str, Any], split: str, seed: int = 3247):
    """The index stream of mix_spectra (data/datasets.py:59-116): np.random.seed(seed); per round
    np.random.choice(range(n_rows), (parallel_samples, n_compounds)), np.unique(axis=0), rows with a
    repeated compound dropped; stops when n * parallel + parallel >= perm(n_rows, n_compounds)."""
    import math
    import numpy as np
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


def mix_spectra(table, idx, ratio, normalize: bool, out_len: int = 1800):
    """The spectrum arithmetic of mix_spectra (data/datasets.py:118-126): np.average(rows, weights=ratio,
    axis=0) in float64, optional normalize_spectrum, zero padding to 1800; float32 at the end (the
    collator's torch.Tensor).  Pinned to the reference's own records: tests/golden/mixture.npz (oracle/make_mixture_goldens.py)."""
    import numpy as np
    table = np.asarray(table)
    out = np.zeros((len(idx), out_len), dtype=np.float32)
    for r, row in enumerate(idx):
        spectra = [table[s].astype(np.float64).tolist() for s in row]
        comb = np.average(spectra, weights=ratio, axis=0).tolist()
        if normalize:
            comb = normalize_spectrum(comb)
        comb = comb + [0] * (out_len - len(comb))
        out[r] = np.asarray(comb, dtype=np.float64).astype(np.float32)
    return out
