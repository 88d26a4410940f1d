"""CPU emulation: logits error of single-pass operand formats against the fp32 oracle at the workload shapes (B = 2).

Every matrix product of the path (linear layers, Q K^T, P V) gets its OPERANDS rounded to the format under test; products and
sums stay fp32 (what an MFMA with fp32 accumulation does).  Residual stream, LayerNorm, softmax statistics fp32.
  python tools/experiments/emul_precision.py c2 fp16 bf16 fp16+head fp16/ffn ...
formats: fp16, bf16, x3 (bf16 hi+lo), `+head` keeps the LM head in x3, `/ffn` `/attn` `/proj` restrict the rounding to that
group of products (the others run x3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import math
import torch
from multimodalanalytical_amd import synth
from multimodalanalytical_amd.params import ParamStore, build_specs
from oracle import afm_oracle as O


def rnd(x, fmt):
    if fmt == "fp16":
        return x.to(torch.float16).to(torch.float32)
    if fmt == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    if fmt == "x3":
        hi = x.to(torch.bfloat16).to(torch.float32)
        return hi + (x - hi).to(torch.bfloat16).to(torch.float32)
    if fmt == "fp32":
        return x
    raise KeyError(fmt)


STATE = {"fmt": "fp32", "groups": None, "head": False, "cur": "proj"}
_lin, _att, _ffn = O.linear, O.attention, O.ffn


def fmt_for(group):
    if STATE["groups"] is None or group in STATE["groups"]:
        return STATE["fmt"]
    return "x3"


def linear(x, w, b):
    f = fmt_for(STATE["cur"])
    if STATE["head"] and w.shape[0] <= 4096 and STATE["cur"] == "head":
        f = "x3"
    y = rnd(x, f) @ rnd(w, f).transpose(-1, -2)
    return y if b is None else y + b


def attention(q, k, v, key_pad, causal):
    f = fmt_for("attn")
    dh = q.shape[-1]
    s = (rnd(q, f) @ rnd(k, f).transpose(-1, -2)) * (1.0 / math.sqrt(dh))
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
    # the kernels round the UNNORMALISED probabilities (<= 1) and divide the output by the fp32 row sum
    o = rnd(e, f) @ rnd(v, f)
    return torch.where(den > 0, o / den.clamp_min(1e-38), torch.zeros_like(o))


def ffn(x, sd, prefix, gated):
    old = STATE["cur"]; STATE["cur"] = "ffn"
    try:
        return _ffn(x, sd, prefix, gated)
    finally:
        STATE["cur"] = old


def model_forward(*a, **k):
    return _mf(*a, **k)


O.linear, O.attention, O.ffn = linear, attention, ffn
_mf = O.model_forward


def patched_model_forward(sd, cfg, data_config, target_modality, enc_inputs, attention_mask, dec_ids, dec_attention_mask,
                          labels=None, memory=None, encoder_align_target=None):
    x = O.embed(sd, data_config, enc_inputs, cfg.get("multimodal_norm", True), cfg["positional_encoding_type"])
    memory = O.encoder(sd, cfg, x, attention_mask)
    dec = O.decoder(sd, cfg, data_config, target_modality, dec_ids, memory, attention_mask, dec_attention_mask)
    STATE["cur"] = "head"
    logits = linear(dec, sd["token_ff.weight"], sd["token_ff.bias"])
    STATE["cur"] = "proj"
    return {"logits": logits}


def main():
    name = sys.argv[1]
    specs = sys.argv[2:] or ["fp16"]
    wl = synth.WORKLOADS[name]
    B = int(os.environ.get("EMUL_B", "2"))
    batch, _ = synth.make_batch(name, B, seed=11)
    inputs = O.batch_to_model_inputs(batch, "Smiles")
    cfg = dict(wl["cfg"], dropout=0.0)
    V = wl["data"]["Smiles"]["vocab_size"]
    ps = ParamStore(build_specs(cfg, wl["data"], V), "cpu", False); ps.init_(5)
    g = torch.Generator().manual_seed(7)
    for s in ps.specs.values():
        if s.kind in ("zeros", "ones"):
            ps.p(s.name).add_(0.05 * torch.randn(s.shape, generator=g))
    sd = {k: v.clone() for k, v in ps.state_dict().items()}
    if cfg["positional_encoding_type"] == "sin_cos":
        sd["embedding.positional_encodings.pos_enc"] = O.sincos_table(cfg["d_model"], cfg["max_position_embeddings"])
    torch.set_num_threads(8)
    enc, am, dec, dm, labels = inputs
    with torch.no_grad():
        STATE.update(fmt="fp32", groups=None, head=False)
        ref = patched_model_forward(sd, cfg, wl["data"], "Smiles", enc, am, dec, dm)["logits"].double()
        top2 = ref.topk(2, dim=-1).values
        margin = (top2[..., 0] - top2[..., 1])
        for spec in specs:
            head = "+head" in spec
            s2 = spec.replace("+head", "")
            groups = None
            if "/" in s2:
                s2, gs = s2.split("/", 1)
                groups = set(gs.split(","))
            STATE.update(fmt=s2, groups=groups, head=head)
            out = patched_model_forward(sd, cfg, wl["data"], "Smiles", enc, am, dec, dm)["logits"].double()
            err = float((out - ref).abs().max() / ref.abs().max())
            flips = int((out.argmax(-1) != ref.argmax(-1)).sum())
            aerr = float((out - ref).abs().max())
            undec = float((margin <= 2 * aerr).double().mean())
            print(f"{name} {spec:20s} logits rel err {err:.2e}  argmax flips {flips} / {ref.shape[0] * ref.shape[1]}  "
                  f"undecidable {undec:.4%}", flush=True)


if __name__ == "__main__":
    main()
