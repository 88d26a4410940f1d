"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from
/root/reference/src) on seeded inputs.  Runs only in the build container (the reference
tree does not exist on the GPU box); the committed .npz files are data: inputs, the
reference's parameters and the reference's outputs.  TEST INFRASTRUCTURE ONLY.

    python oracle/make_goldens.py            # rewrites tests/golden/

What is captured:
  model_plain.npz / model_gated_learned.npz
      cfg + data_config (json), state_dict, 4 micro-batches (seq-first batch dicts as the
      collator emits them, datamodules.py:201-218), per micro-batch: logits, loss,
      teacher-forced argmax, token accuracy; gradients of micro-batch 0; parameters after
      1 and 2 optimiser steps of accumulate-4 / clip 1.0 / AdamW + OneCycleLR driven by
      torch.optim + torch.nn.utils.clip_grad_norm_ (what Lightning calls); greedy ids.
  embed_variants.npz   linear_2_layer / linear_3_layer / msms_number / xVal embedders.
  schedule.npz         OneCycleLR lr and beta1 for total_steps=100, sin-cos table rows.
  patches.npz          PatchPreprocessor (data/preprocessing/patches.py) outputs on seeded synthetic
                       spectra: plain, interpolated, overlapping, derivative, masking, None rows.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference/src")
from analytical_fm.modeling.custom_modeling import AlignConfig, CustomConfig, CustomModel  # noqa: E402
from analytical_fm.modeling.utils import MultimodalEmbedding  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
SEED = 3247  # configuration.py:10


class Tok:
    def __init__(self, v):
        self.vocab_size = v
        self.pad_token_id, self.bos_token_id, self.eos_token_id = 0, 2, 3


def build(cfg_kwargs, data_config, target="Smiles"):
    torch.manual_seed(SEED)
    kw = dict(cfg_kwargs)
    if kw.get("align_config"):   # load_custom_model converts the yaml dict (wrapper.py:164-165)
        kw["align_config"] = AlignConfig(**kw["align_config"])
    cfg = CustomConfig(**kw)
    emb = MultimodalEmbedding(data_config, cfg.d_model, True, do_positional_encodings=True,
                              positional_encodings_type=cfg.positional_encoding_type,
                              max_seq_len=cfg.max_position_embeddings)
    model = CustomModel(target, Tok(data_config[target]["vocab_size"]), cfg, emb)
    # HFWrapper._init_params (wrapper.py:320-327): xavier on every param with dim > 1
    for p in model.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_uniform_(p)
    # make biases / LN affine non-trivial so goldens exercise them
    g = torch.Generator().manual_seed(SEED + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                if "norm" in n and n.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return cfg, emb, model


def make_batch(rng, data_config, B, lens, T, full_mask_sample=None):
    """Seq-first batch dict exactly as MultiModalDataCollator emits it."""
    enc_in, masks = {}, []
    for m, mc in data_config.items():
        if mc["target"]:
            continue
        L = lens[m]
        if mc["type"] in ("text", "multiplets", "carbon"):
            V = mc["vocab_size"]
            ids = np.zeros((L, B), dtype=np.int64)
            pad = np.ones((L, B), dtype=bool)
            for b in range(B):
                n = int(rng.integers(max(3, L // 2), L + 1))
                if full_mask_sample is not None and full_mask_sample == (m, b):
                    n = 0  # a `None` sample: fully masked row (multiplets.py)
                if n:
                    ids[:n, b] = np.concatenate([[2], rng.integers(4, V, size=n - 2), [3]])
                    pad[:n, b] = False
            enc_in[m] = torch.from_numpy(ids)
            masks.append(torch.from_numpy(pad))
        else:
            ps = mc["preprocessor_arguments"]["patch_size"]
            x = rng.standard_normal((L, B, ps)).astype(np.float32)
            enc_in[m] = torch.from_numpy(x)
            masks.append(torch.zeros((L, B), dtype=torch.bool))
    tgt_V = [mc for mc in data_config.values() if mc["target"]][0]["vocab_size"]
    ids = np.zeros((T + 1, B), dtype=np.int64)
    for b in range(B):
        n = int(rng.integers(T // 2, T + 2))
        ids[:n, b] = np.concatenate([[2], rng.integers(4, tgt_V, size=n - 2), [3]])
    ids = torch.from_numpy(ids)
    return {
        "encoder_input": enc_in,
        "encoder_pad_mask": torch.cat(masks, dim=0),
        "decoder_input": {"Smiles": ids[:-1]},
        "decoder_pad_mask": ids[:-1] == 0,
        "target": ids[1:].clone(),
    }


def wrapper_forward(model, emb, batch, train=True):
    """HFWrapper.forward (wrapper.py:346-407) without Lightning."""
    input_ids = {m: v.transpose(1, 0) for m, v in batch["encoder_input"].items()}
    dec_in = batch["decoder_input"]["Smiles"].transpose(1, 0)
    am = (~batch["encoder_pad_mask"]).int().T
    dm = (~batch["decoder_pad_mask"]).int().T
    labels = batch["target"].T.contiguous().clone()
    labels[labels == 0] = -100
    model.train(train)
    kwargs = {}
    if "encoder_alignment_input" in batch:      # wrapper.py:395-396
        kwargs["encoder_align_target"] = batch["encoder_alignment_input"]
    out = model(inputs_embeds=emb(input_ids), attention_mask=am, decoder_input_ids=dec_in,
                decoder_attention_mask=dm, labels=labels, **kwargs)
    return out


def token_acc(batch, logits):  # wrapper.py:641-655
    tok = batch["target"].T
    pred = torch.argmax(logits, dim=-1)
    mask = tok != -100
    return ((tok == pred) * mask).sum().float() / mask.sum().float()


def greedy(model, emb, batch, max_length):
    """Own loop over the reference's `generating` branch (custom_modeling.py:447-455)."""
    model.eval()
    torch.backends.mha.set_fastpath_enabled(False)
    with torch.no_grad():
        input_ids = {m: v.transpose(1, 0) for m, v in batch["encoder_input"].items()}
        am = (~batch["encoder_pad_mask"]).int().T
        enc = model.encoder(emb(input_ids), attention_mask=am)
        B = am.shape[0]
        ids = torch.full((B, 1), 2, dtype=torch.long)
        done = torch.zeros(B, dtype=torch.bool)
        while ids.shape[1] < max_length:
            lg = model(encoder_outputs=dict(enc), attention_mask=am, decoder_input_ids=ids).logits
            nxt = lg[:, -1].argmax(-1)
            if ids.shape[1] == max_length - 1:
                nxt = torch.full_like(nxt, 3)
            nxt = torch.where(done, torch.zeros_like(nxt), nxt)
            ids = torch.cat([ids, nxt[:, None]], 1)
            done |= nxt == 3
            if bool(done.all()):
                break
    model.train()
    return ids


def dump_model_case(name, cfg_kwargs, data_config, lens, T, B=4, full_mask=None, lr=1e-3,
                    total_steps=10):
    cfg, emb, model = build(cfg_kwargs, data_config)
    rng = np.random.default_rng(SEED)
    out = {}
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()
          if not k.startswith("decoder.embedding.")}  # alias of embedding.*
    for k, v in sd.items():
        out[f"sd/{k}"] = v.numpy()
    batches = [make_batch(rng, data_config, B, lens, T, full_mask if i == 1 else None)
               for i in range(4)]
    if cfg_kwargs.get("align_config"):
        n_out = cfg_kwargs["align_config"]["output_dimension"]
        for b in batches:            # min-max normalised target spectra: values in [0, 1] with exact zeros
            t = rng.random((B, n_out)).astype(np.float32)
            t[:, ::7] = 0.0
            b["encoder_alignment_input"] = torch.from_numpy(t)
    for i, b in enumerate(batches):
        for m, v in b["encoder_input"].items():
            out[f"b{i}/encoder_input/{m}"] = v.numpy()
        out[f"b{i}/encoder_pad_mask"] = b["encoder_pad_mask"].numpy()
        out[f"b{i}/decoder_input/Smiles"] = b["decoder_input"]["Smiles"].numpy()
        out[f"b{i}/decoder_pad_mask"] = b["decoder_pad_mask"].numpy()
        out[f"b{i}/target"] = b["target"].numpy()
        if "encoder_alignment_input" in b:
            out[f"b{i}/encoder_alignment_input"] = b["encoder_alignment_input"].numpy()
    # forward / backward per micro-batch (train mode, dropout 0)
    for i, b in enumerate(batches):
        model.zero_grad()
        o = wrapper_forward(model, emb, b)
        o.loss.backward()
        out[f"b{i}/logits"] = o.logits.detach().numpy()
        out[f"b{i}/loss"] = o.loss.detach().numpy()
        out[f"b{i}/argmax"] = o.logits.argmax(-1).numpy()
        out[f"b{i}/token_acc"] = token_acc(b, o.logits).numpy()
        out[f"b{i}/encoder_hidden_states"] = o.encoder_hidden_states.detach().numpy()
        if cfg_kwargs.get("align_config"):
            out[f"b{i}/model_only_loss"] = o.loss_dict["model_only_loss"].detach().numpy()
            out[f"b{i}/alignment_loss"] = o.loss_dict["alignment_loss"].detach().numpy()
        if i == 0:
            for n, p in model.named_parameters():
                if n.startswith("decoder.embedding."):
                    continue
                out[f"grad0/{n}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    # optimiser: accumulate 4, clip 1.0, AdamW + OneCycle (torch.optim drives it here)
    model.zero_grad()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=0.01, betas=(0.9, 0.999))
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, lr, total_steps=total_steps)
    for step in range(2):
        for b in batches:
            o = wrapper_forward(model, emb, b)
            (o.loss / 4).backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        out[f"step{step + 1}/grad_norm"] = norm.numpy()
        opt.step(); sch.step(); opt.zero_grad()
        for n, p in model.named_parameters():
            if n.startswith("decoder.embedding."):
                continue
            out[f"step{step + 1}/{n}"] = p.detach().clone().numpy()
    # greedy decode from the INITIAL weights
    model.load_state_dict({**model.state_dict(), **sd})
    out["greedy/ids"] = greedy(model, emb, batches[0], max_length=12).numpy()
    meta = {"cfg": cfg_kwargs, "data_config": data_config, "lens": lens, "T": T, "B": B,
            "lr": lr, "total_steps": total_steps, "weight_decay": 0.01, "optimiser": "adamw",
            "acc_batches": 4, "clip": 1.0, "greedy_max_length": 12}
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "loss", [float(out[f"b{i}/loss"]) for i in range(4)],
          "bytes", os.path.getsize(os.path.join(OUT, name + ".npz")))


def dump_embed_variants():
    torch.manual_seed(SEED)
    dc = {
        "A": {"type": "1D_patches", "target": False,
              "preprocessor_arguments": {"patch_size": 10, "encoding_type": "linear_2_layer"}},
        "B": {"type": "1D_patches", "target": False,
              "preprocessor_arguments": {"patch_size": 6, "encoding_type": "linear_3_layer"}},
        "C": {"type": "msms_number", "target": False, "preprocessor_arguments": {}},
        "D": {"type": "multiplets", "vocab_size": 30, "pad_token_id": 0, "target": False},
        "Smiles": {"type": "text", "vocab_size": 26, "pad_token_id": 0, "target": True},
    }
    out = {}
    for pe in ("sin_cos", "learned"):
        emb = MultimodalEmbedding(dc, 48, True, do_positional_encodings=True,
                                  positional_encodings_type=pe, max_seq_len=64)
        for p in emb.parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_uniform_(p)
        B = 3
        inp = {"A": torch.randn(B, 5, 10), "B": torch.randn(B, 4, 6), "C": torch.rand(B, 7, 2),
               "D": {"tokenized_input": torch.randint(1, 30, (B, 6)),
                     "numerical_values": torch.rand(B, 6) * 3}}
        y = emb(inp)
        for k, v in emb.state_dict().items():
            out[f"{pe}/sd/embedding.{k}"] = v.numpy()
        out[f"{pe}/in/A"], out[f"{pe}/in/B"], out[f"{pe}/in/C"] = (inp[k].numpy() for k in "ABC")
        out[f"{pe}/in/D/tokenized_input"] = inp["D"]["tokenized_input"].numpy()
        out[f"{pe}/in/D/numerical_values"] = inp["D"]["numerical_values"].numpy()
        out[f"{pe}/out"] = y.detach().numpy()
    out["meta"] = np.frombuffer(json.dumps({"data_config": dc, "d_model": 48}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "embed_variants.npz"), **out)


def dump_patches():
    """G6 (data side): the reference's PatchPreprocessor on synthetic spectra."""
    from analytical_fm.data.preprocessing.patches import PatchPreprocessor
    rng = np.random.default_rng(SEED + 7)
    out = {}
    cases = [
        ("ps125", dict(patch_size=125, masking=False, interpolation=False), 1800, 6),
        ("ps75_interp", dict(patch_size=75, masking=False, interpolation=True), 1800, 5),
        ("ps75_interp1791", dict(patch_size=75, masking=False, interpolation=True), 1791, 3),
        ("ps2", dict(patch_size=2, masking=False, interpolation=False), 1984, 4),
        ("ps50_overlap2", dict(patch_size=50, masking=False, interpolation=False, overlap=2), 1800, 4),
        ("ps100_deriv", dict(patch_size=100, masking=False, interpolation=False, derivative=True), 1800, 4),
        ("ps125_masking", dict(patch_size=125, masking=True, interpolation=False), 1800, 5),
    ]
    meta = {}
    for name, kw, L, B in cases:
        pp = PatchPreprocessor(**kw)
        # fit statistics the way initialise() does (mean/std over non-zero entries, patches.py:37-39)
        fit = np.abs(rng.standard_normal((16, L))).astype(np.float32).astype(np.float64)
        fit[:, :7] = 0.0
        pp.mean = fit[fit != 0].mean()
        pp.std = fit[fit != 0].std()
        spectra = np.abs(rng.standard_normal((B, L))).astype(np.float32)
        present = np.ones(B, dtype=bool)
        if B >= 4 and "interp" not in name:      # interpolate() cannot take the zero-filled None rows of another length
            present[2] = False
        if name == "ps125_masking":                # a patch that standardises to exact zeros is masked
            spectra[0, 125:250] = np.float32(pp.mean)
        rows = [spectra[i].astype(np.float64).tolist() if present[i] else None for i in range(B)]
        patches, mask = pp(rows)
        out[f"{name}/spectra"] = spectra
        out[f"{name}/present"] = present
        out[f"{name}/patches"] = patches.numpy()
        out[f"{name}/mask"] = mask.numpy()
        meta[name] = dict(kw, mean=float(pp.mean), std=float(pp.std), L=L)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "patches.npz"), **out)
    print("patches.npz", {k: v.shape for k, v in out.items() if k.endswith("patches")})


def dump_schedule():
    from analytical_fm.modeling.utils import SincCosPositionalEncoding
    p = torch.nn.Parameter(torch.zeros(1))
    out = {}
    for total in (10, 100):
        opt = torch.optim.AdamW([p], lr=1e-3, betas=(0.9, 0.999))
        sch = torch.optim.lr_scheduler.OneCycleLR(opt, 1e-3, total_steps=total)
        lrs, b1s = [], []
        for _ in range(total):
            lrs.append(opt.param_groups[0]["lr"]); b1s.append(opt.param_groups[0]["betas"][0])
            opt.step(); sch.step() if len(lrs) < total else None
        out[f"onecycle{total}/lr"], out[f"onecycle{total}/beta1"] = np.array(lrs), np.array(b1s)
    for d in (64, 128, 30):
        out[f"sincos/{d}"] = SincCosPositionalEncoding(d, 40).pos_enc.numpy()
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--only-patches" in sys.argv:
        dump_patches()
        sys.exit(0)
    ONLY_ALIGN = "--only-align" in sys.argv
    ONLY_OPTIONS = "--only-options" in sys.argv     # round 3: post-LN / ReLU layer options (configs/model/*.yaml surface)
    dc_plain = {
        "Formula": {"type": "text", "vocab_size": 45, "pad_token_id": 0, "target": False},
        "IR": {"type": "1D_patches", "target": False,
               "preprocessor_arguments": {"patch_size": 125, "interpolation": False, "masking": False}},
        "Smiles": {"type": "text", "vocab_size": 26, "pad_token_id": 0, "target": True},
    }
    base = dict(d_model=64, max_position_embeddings=128, encoder_layers=2, decoder_layers=2,
                encoder_attention_heads=4, decoder_attention_heads=4, encoder_ffn_dim=128,
                decoder_ffn_dim=128, dropout=0.0)
    if ONLY_OPTIONS:
        # post_layer_normalisation=False is torch's norm_first=False (custom_modeling.py:129,176); activation_function goes straight to
        # nn.TransformerEncoderLayer / DecoderLayer (custom_modeling.py:127,174)
        dump_model_case("model_postln_relu", dict(base, post_layer_normalisation=False, activation_function="relu"), dc_plain,
                        {"Formula": 10, "IR": 14}, T=20)
        dump_model_case("model_postln_gated", dict(base, post_layer_normalisation=False, gated_linear=True,
                                                   positional_encoding_type="learned"), dc_plain, {"Formula": 10, "IR": 14}, T=20,
                        full_mask=None)
        sys.exit(0)
    if not ONLY_ALIGN:
        dump_model_case("model_plain", dict(base), dc_plain, {"Formula": 10, "IR": 14}, T=20)
    dc_multi = {
        "Formula": {"type": "text", "vocab_size": 45, "pad_token_id": 0, "target": False},
        "IR": {"type": "1D_patches", "target": False,
               "preprocessor_arguments": {"patch_size": 75, "interpolation": False, "masking": False}},
        "Multiplets": {"type": "multiplets", "vocab_size": 60, "pad_token_id": 0, "target": False},
        "Carbon": {"type": "carbon", "vocab_size": 50, "pad_token_id": 0, "target": False},
        "Smiles": {"type": "text", "vocab_size": 26, "pad_token_id": 0, "target": True},
    }
    if not ONLY_ALIGN:
        dump_model_case("model_gated_learned",
                        dict(base, gated_linear=True, positional_encoding_type="learned"),
                        dc_multi, {"Formula": 8, "IR": 6, "Multiplets": 17, "Carbon": 9}, T=16,
                        full_mask=("Multiplets", 2))
    al = dict(hidden_dimension=32, conv_channels=8, kernel_size=3, output_dimension=40, loss_lambda=0.5)
    dc_small = {k: dc_plain[k] for k in ("Formula", "IR", "Smiles")}
    small = dict(base, encoder_layers=1, decoder_layers=1)
    dump_model_case("model_align_mlp_mse", dict(small, align_config=dict(al, align_network="mlp", loss_function="mse")),
                    dc_small, {"Formula": 10, "IR": 14}, T=12)
    dump_model_case("model_align_conv_sid", dict(small, align_config=dict(al, align_network="convolutional", loss_function="sid")),
                    dc_small, {"Formula": 10, "IR": 14}, T=12, full_mask=None)
    dump_model_case("model_align_mlp_mae", dict(small, align_config=dict(al, align_network="mlp", loss_function="mae")),
                    dc_small, {"Formula": 10, "IR": 14}, T=12)
    if ONLY_ALIGN:
        sys.exit(0)
    dump_embed_variants()
    dump_schedule()
    dump_patches()
