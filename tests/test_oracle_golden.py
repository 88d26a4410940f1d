"""CPU: the oracle (oracle/afm_oracle.py) against vectors produced by the reference itself
(tests/golden/*.npz, generator oracle/make_goldens.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import afm_oracle as O
from tests import golden_io as G

CASES = ["model_plain", "model_gated_learned", "model_align_mlp_mse", "model_align_conv_sid", "model_align_mlp_mae",
         "model_postln_relu", "model_postln_gated"]      # (the last two: post_layer_normalisation=False, activation "relu")


@pytest.fixture(scope="module", params=CASES)
def case(request):
    t = G.load(request.param)
    return t, G.model_cfg(t["meta"])


def _fwd(t, cfg, i, sd=None):
    b = G.batch_of(t, i)
    enc, am, dec, dm, labels = O.batch_to_model_inputs(b, "Smiles")
    return O.model_forward(sd or t["sd"], cfg, t["meta"]["data_config"], "Smiles", enc, am, dec, dm, labels,
                           encoder_align_target=b.get("encoder_alignment_input"))


def test_forward_logits_loss_argmax(case):
    t, cfg = case
    for i in range(4):
        out = _fwd(t, cfg, i)
        ref = t[f"b{i}"]
        torch.testing.assert_close(out["logits"], ref["logits"], rtol=1e-5, atol=2e-6)
        torch.testing.assert_close(out["loss"], ref["loss"], rtol=1e-6, atol=1e-6)
        assert torch.equal(out["logits"].argmax(-1), ref["argmax"])  # bit-exact token ids
        torch.testing.assert_close(out["encoder_hidden_states"], ref["encoder_hidden_states"], rtol=1e-5, atol=2e-6)
        acc = O.token_accuracy(out["logits"], ref["target"].T)
        torch.testing.assert_close(acc, ref["token_acc"])
        if cfg.get("align_config"):     # loss_dict of CustomLMOutput (custom_modeling.py:494-497)
            torch.testing.assert_close(out["loss_dict"]["alignment_loss"], ref["alignment_loss"], rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(out["loss_dict"]["model_only_loss"], ref["model_only_loss"], rtol=1e-6, atol=1e-6)


def test_backward_grads(case):
    t, cfg = case
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith("positional_encodings.pos_enc")) for k, v in t["sd"].items()}
    out = _fwd(t, cfg, 0, sd)
    out["loss"].backward()
    for k, g in t["grad0"].items():
        got = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        torch.testing.assert_close(got, g, rtol=2e-4, atol=2e-7, msg=lambda m: f"{k}: {m}")


def test_two_optimizer_steps(case):
    t, cfg = case
    m = t["meta"]
    tr = O.OracleTrainer(t["sd"], cfg, m["data_config"], "Smiles", lr=m["lr"], total_steps=m["total_steps"],
                         optimiser=m["optimiser"], weight_decay=m["weight_decay"], acc_batches=m["acc_batches"],
                         clip=m["clip"])
    for step in (1, 2):
        for i in range(4):
            b = G.batch_of(t, i)
            tr.micro_batch(*O.batch_to_model_inputs(b, "Smiles"), encoder_align_target=b.get("encoder_alignment_input"))
        torch.testing.assert_close(tr.last_norm, t[f"step{step}"]["grad_norm"], rtol=1e-5, atol=1e-6)
        for k, ref in t[f"step{step}"].items():
            if k == "grad_norm":
                continue
            got = tr.sd[k].detach()
            if k.endswith("in_proj_bias"):
                # the K-bias gradient is identically zero in exact arithmetic (softmax shift
                # invariance); Adam normalises its rounding noise to O(lr) moves, so that third
                # is only bounded by the summed learning rates.
                d = got.numel() // 3
                torch.testing.assert_close(got[d:2 * d], ref[d:2 * d], rtol=0, atol=3e-4)
                got, ref = torch.cat([got[:d], got[2 * d:]]), torch.cat([ref[:d], ref[2 * d:]])
            torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-6, msg=lambda s: f"step{step} {k}: {s}")


def test_greedy_decode(case):
    t, cfg = case
    enc, am, _, _, _ = O.batch_to_model_inputs(G.batch_of(t, 0), "Smiles")
    ids = O.greedy_decode(t["sd"], cfg, t["meta"]["data_config"], "Smiles", enc, am,
                          max_length=t["meta"]["greedy_max_length"])
    assert torch.equal(ids, t["greedy"]["ids"])


def test_embed_variants():
    t = G.load("embed_variants")
    dc = t["meta"]["data_config"]
    for pe in ("sin_cos", "learned"):
        inp = {"A": t[pe]["in"]["A"], "B": t[pe]["in"]["B"], "C": t[pe]["in"]["C"], "D": dict(t[pe]["in"]["D"])}
        y = O.embed(t[pe]["sd"], dc, inp, True, pe)
        torch.testing.assert_close(y, t[pe]["out"], rtol=1e-5, atol=2e-6)


def test_schedule_and_sincos():
    t = G.load("schedule")
    for total in (10, 100):
        lr = [O.onecycle(s, total, 1e-3)[0] for s in range(total)]
        b1 = [O.onecycle(s, total, 1e-3)[1] for s in range(total)]
        np.testing.assert_allclose(lr, t[f"onecycle{total}"]["lr"].numpy(), rtol=1e-12)
        np.testing.assert_allclose(b1, t[f"onecycle{total}"]["beta1"].numpy(), rtol=1e-12)
    for d in (64, 128, 30):
        torch.testing.assert_close(O.sincos_table(d, 40), t["sincos"][str(d)], rtol=0, atol=1e-6)


def test_patch_preprocessor_oracle_matches_reference_bit_exact():
    """oracle.patch_preprocess vs PatchPreprocessor.__call__ outputs captured from the reference
    (tests/golden/patches.npz, oracle/make_goldens.py:dump_patches): plain, interpolated (both source
    lengths), patch size 2, overlapping, derivative, masking, `None` rows.  Bit-exact fp32 + masks."""
    import json
    import numpy as np
    from oracle import afm_oracle as O
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "patches.npz"))
    meta = json.loads(bytes(g["meta"]).decode())
    assert len(meta) == 7
    for name, kw in meta.items():
        p, m = O.patch_preprocess(g[f"{name}/spectra"], g[f"{name}/present"], kw["mean"], kw["std"], kw["patch_size"],
                                  kw["masking"], kw["interpolation"], kw.get("overlap", 1), kw.get("derivative", False))
        assert p.dtype == np.float32 and np.array_equal(p, g[f"{name}/patches"]), name
        assert np.array_equal(m, g[f"{name}/mask"]), name


@pytest.mark.parametrize("name", ["model_plain", "model_postln_relu"])
def test_stock_torch_wiring_equals_oracle_and_reference_golden(name):
    """oracle/stock_torch.py (nn.TransformerEncoder / Decoder as the reference wires them, SURVEY 8c item 2: the
    timed "reference PyTorch CPU path" of bench.py) reproduces the reference's own logits / loss / gradients
    (also with the post-LN / ReLU layer options)."""
    from oracle import stock_torch as ST
    t = G.load(name); cfg = dict(G.model_cfg(t["meta"]), dropout=0.0)
    dc = t["meta"]["data_config"]
    m = ST.StockSeq2Seq(cfg, dc["Smiles"]["vocab_size"])
    m.load_oracle_state(t["sd"])
    m.train()
    enc, am, dec, dm, labels = O.batch_to_model_inputs(G.batch_of(t, 0), "Smiles")
    x_enc, x_dec = ST.embed_inputs(t["sd"], cfg, dc, "Smiles", enc, dec)
    logits, loss = m(x_enc, am, x_dec, dm, labels)
    ref = t["b0"]
    torch.testing.assert_close(logits, ref["logits"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(loss, ref["loss"], rtol=1e-5, atol=1e-5)
    loss.backward()
    for k in ("encoder.layers.0.linear1.weight", "decoder.layers.1.multihead_attn.out_proj.weight", "token_ff.weight"):
        g = dict(m.named_parameters())[k].grad
        torch.testing.assert_close(g, t["grad0"][k], rtol=2e-4, atol=2e-6, msg=lambda s: f"{k}: {s}")


# ------------------------------------------------------------------ mixture generator (data/datasets.py:49-141)
def _mixture_cases():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mixture.npz"))
    for tag in z["cases"]:
        tag = str(tag)
        ratio = z[f"{tag}/cfg_ratio"].tolist() or None
        cfg = dict(n_compounds=int(z[f"{tag}/cfg_n_compounds"]), compounds_ratio=ratio, parallel_samples=int(z[f"{tag}/cfg_parallel"]),
                   train_max_n_samples=int(z[f"{tag}/cfg_max_n"]), normalize=bool(z[f"{tag}/cfg_normalize"]), mixed=bool(z[f"{tag}/cfg_mixed"]))
        yield tag, cfg, {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(tag + "/")}
    return


def mixture_records_from(index_rounds, mix_fn, table, cfg):
    """The reference's record stream rebuilt from an index stream and a mixing function: for every index row, one record per
    compound with a non-zero ratio (data/datasets.py:107-141)."""
    nc = cfg["n_compounds"]
    ratio = cfg["compounds_ratio"] or [1 / nc] * nc
    ir, tgt, smi, add, pct = [], [], [], [], []
    for ri in index_rounds:
        mixed = mix_fn(table, ri, ratio, cfg["normalize"])
        for r, row in enumerate(ri):
            for i in range(nc):
                if ratio[i] == 0:
                    continue
                ir.append(mixed[r]); tgt.append(table[row[i]]); smi.append(int(row[i]))
                add.append(",".join(f"S{row[j]}" for j in range(nc) if j != i)); pct.append(f"{ratio[i]}")
    return np.asarray(ir), np.asarray(tgt), np.asarray(smi), np.asarray(add), np.asarray(pct)


def test_mixture_generator_restatement_reproduces_the_reference_records():
    """oracle.mix_indices + oracle.mix_spectra (and the package's host-side index stream) against the records the reference's own
    mix_spectra / normalize_spectrum produced (tests/golden/mixture.npz, oracle/make_mixture_goldens.py): which rows are mixed, in
    which order, the mixed spectrum to the bit (as the float32 the collator makes of it), targets, partner names, percentages."""
    from multimodalanalytical_amd.preprocess import mix_indices
    seen = 0
    for tag, cfg, g in _mixture_cases():
        table = g["table"]
        if cfg["mixed"]:
            want = np.asarray([O.normalize_spectrum(r.astype(np.float64).tolist()) if cfg["normalize"] else r for r in table])
            assert np.array_equal(want, g["ir"]) and np.array_equal(g["smiles"], np.arange(len(table)))
            assert not g["ir_target"].any() and set(g["percentage"]) == {f"{1 / cfg['n_compounds']}"}
            continue
        for stream in (O.mix_indices, mix_indices):
            rounds = list(stream(len(table), cfg, "train", seed=3247))
            ir, tgt, smi, add, pct = mixture_records_from(rounds, O.mix_spectra, table, cfg)
            assert len(smi) == len(g["smiles"]) > 0, tag
            assert np.array_equal(smi, g["smiles"]), tag
            assert np.array_equal(ir, g["ir"].astype(np.float32)), tag
            assert np.array_equal(tgt.astype(np.float64), g["ir_target"]), tag
            assert list(add) == list(g["additional"]) and list(pct) == list(g["percentage"]), tag
        seen += 1
    assert seen >= 5
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mixture.npz"))
    for i in range(3):
        assert O.normalize_spectrum(z[f"normalize/{i}/in"].tolist()) == z[f"normalize/{i}/out"].tolist()
