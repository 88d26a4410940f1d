"""GPU: BASELINE.json's workloads c2..c5 at their REAL layer shapes (d512/h8/f2048 and d768/h12/f3072 gated +
learned positions, S = 1024 with the synthetic set's pad masks, T = 128 / 256), batch 2 so the CPU oracle
finishes in seconds: forward AND backward of the HIP engine against the oracle, every precision mode.

Bars (max |logit - ref| / max |ref|; gradients by relative norm per parameter tensor):
  fp32    exact-fp32 FMA kernels                 logits 1e-4, ids bit-exact, grads 2e-3
  bf16x3  split bf16 pairs, 3 MFMAs / product    logits 1e-3 (north star), ids bit-exact under the margin
          policy below, grads 5e-3
  bf16x3-mixed  bf16x3 forward, single-pass bf16 backward on the hi planes of the saved pair tensors: logits and ids
          exactly as bf16x3 (same kernels), grads at the bf16 bar 8e-2 per tensor and 1e-2 over all parameters
  bf16    single bf16 MFMA pass                  logits 3e-2, ids exact outside twice the measured error, grads 8e-2
  fp16    single fp16 MFMA pass forward and backward + loss scaling (the reference's GPU precision, "16-mixed"):
          logits 1e-3 (north star; measured 4e-4 .. 7e-4), ids exact outside twice the measured error, grads 2e-2 per
          tensor and 3e-3 over all parameters (the gradient buffer holds S x the gradients: divided out here)
Margin policy for "bit-exact argmax": a position whose two largest REFERENCE logits are closer than twice the
MEASURED maximum logit error cannot be decided by the arithmetic under test (nor by the reference run on another
BLAS); there the id must be a candidate whose reference logit lies within that band of the maximum.  Everywhere else ids
must be equal.  bf16x3 measures ~1e-5, so
at most a handful of exact near-ties are undecidable (>= 99.9 % of the positions must be decidable); fp32 mode
is held to plain equality.
"""
import functools

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import afm_oracle as O  # noqa: E402

DEV = "cuda:0"
MODES = ["fp32", "bf16x3", "bf16x3-mixed", "bf16", "fp16"]


def _dtype(mode):
    from multimodalanalytical_amd.x2 import X2
    return {"fp32": torch.float32, "bf16": torch.bfloat16, "bf16x3": X2.dtype, "bf16x3-mixed": X2.dtype, "fp16": torch.float16}[mode]


def _bdtype(mode):
    return torch.bfloat16 if mode == "bf16x3-mixed" else None


@functools.lru_cache(maxsize=None)
def _case(name):
    """(inputs, state dict, oracle logits / loss / gradients) of workload `name`, B = 2, dropout 0."""
    from multimodalanalytical_amd import synth
    from multimodalanalytical_amd.params import ParamStore, build_specs
    wl = synth.WORKLOADS[name]
    batch, _ = synth.make_batch(name, 2, seed=11)
    inputs = O.batch_to_model_inputs(batch, "Smiles")
    cfg = dict(wl["cfg"], dropout=0.0)
    V = wl["data"]["Smiles"]["vocab_size"]
    ps = ParamStore(build_specs(cfg, wl["data"], V), "cpu", False)
    ps.init_(5)
    # non-trivial biases / LayerNorm parameters so every gradient path carries signal
    g = torch.Generator().manual_seed(7)
    for s in ps.specs.values():
        if s.kind in ("zeros", "ones"):
            ps.p(s.name).add_(0.05 * torch.randn(s.shape, generator=g))
    sd = {k: v.clone() for k, v in ps.state_dict().items()}
    if cfg["positional_encoding_type"] == "sin_cos":
        sd["embedding.positional_encodings.pos_enc"] = O.sincos_table(cfg["d_model"], cfg["max_position_embeddings"])
    torch.set_num_threads(min(32, torch.get_num_threads()))
    leaf = {k: v.clone().requires_grad_(not k.endswith("pos_enc")) for k, v in sd.items()}
    ref = O.model_forward(leaf, cfg, wl["data"], "Smiles", *inputs)
    ref["loss"].backward()
    grads = {k: v.grad.detach() for k, v in leaf.items() if v.grad is not None}
    return wl, cfg, inputs, sd, {"logits": ref["logits"].detach(), "loss": ref["loss"].detach()}, grads


def _to(x):
    return {k: _to(v) for k, v in x.items()} if isinstance(x, dict) else x.to(DEV)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5"])
def test_shape_parity_forward_backward_vs_oracle(name, mode):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodalanalytical_amd import ops
    from multimodalanalytical_amd.engine import Seq2SeqEngine
    wl, cfg, inputs, sd, ref, grads = _case(name)
    eng = Seq2SeqEngine(cfg, wl["data"], "Smiles", wl["data"]["Smiles"]["vocab_size"], device=DEV,
                        compute_dtype=_dtype(mode), seed=5, backward_dtype=_bdtype(mode))
    eng.load_state_dict(sd)
    enc, am, dec, dm, labels = inputs
    ops.reset_algo_log()
    out = eng.forward(_to(enc), am.to(DEV), dec.to(DEV), dm.to(DEV), labels.to(DEV), backward=True)
    algos = ops.algo_log()
    if mode != "fp32":   # the MFMA kernels really ran (not the exact-fp32 FMA kernels; c1 has 16-wide heads: FMA attention)
        assert name == "c1" or any(a.startswith("attn_mfma") for a in algos), algos
        assert any(a.startswith("mfma_nt") for a in algos), algos
        assert name == "c1" or any(a.startswith("mfma_tn") for a in algos), algos      # (c1: 320 token rows, below the MFMA wgrad's K)
        if cfg["gated_linear"]:   # c4 / c5: the gated FFN runs through the fused GLU epilogues
            assert any("glu" in a for a in algos), algos
    logits, rl = out["logits"].cpu().double(), ref["logits"].double()
    scale = float(rl.abs().max())
    err = float((logits - rl).abs().max()) / scale
    tol = {"fp32": 1e-4, "bf16x3": 1e-3, "bf16x3-mixed": 1e-3, "bf16": 3e-2, "fp16": 1e-3}[mode]
    assert err < tol, (name, mode, err)
    ids, rid = out["argmax"].cpu(), ref["logits"].argmax(-1)
    top2 = ref["logits"].topk(2, -1)
    margin = (top2.values[..., 0] - top2.values[..., 1]).double()
    if mode == "fp32":
        assert torch.equal(ids, rid)
    else:
        band = 2 * err * scale
        sure = margin > band
        assert torch.equal(ids[sure], rid[sure]), (name, mode)
        # inside the band any candidate whose REFERENCE logit is within the band of the maximum may win (near-ties of 2+ ids)
        chosen = ref["logits"].double().gather(-1, ids.unsqueeze(-1)).squeeze(-1)
        assert bool((chosen >= top2.values[..., 0].double() - band).all())
        # measured (gpurun_out/parity_records.jsonl, round 4): fp16 leaves 2 .. 5 of 256 / 4 of 512 positions undecidable on these
        # near-flat fresh-init logits, single-pass bf16 21 .. 38 of 256
        assert float(sure.double().mean()) > (0.999 if mode.startswith("bf16x3") else 0.97 if mode == "fp16" else 0.8)
    ltol = {"fp32": 1e-5, "bf16x3": 1e-4, "bf16x3-mixed": 1e-4, "bf16": 2e-2, "fp16": 2e-3}[mode]
    torch.testing.assert_close(out["loss"].cpu(), ref["loss"], rtol=ltol, atol=ltol)
    gtol = {"fp32": 2e-3, "bf16x3": 5e-3, "bf16x3-mixed": 8e-2, "bf16": 8e-2, "fp16": 2e-2}[mode]
    gmax = max(float(g.norm()) for g in grads.values())
    S = float(eng.scaler[0]) if eng.scaler is not None else 1.0      # fp16: the buffer holds S x the gradients
    assert (S > 1.0) == (mode == "fp16")
    if mode in ("bf16x3-mixed", "fp16"):   # all parameters together: 16-bit backward on parity-grade / fp16 activations
        num = sum(float((eng.ps.g(k).cpu() / S - g).norm()) ** 2 for k, g in grads.items() if not k.endswith("in_proj_bias"))
        den = sum(float(g.norm()) ** 2 for k, g in grads.items() if not k.endswith("in_proj_bias"))
        assert (num / den) ** 0.5 < (1e-2 if mode == "bf16x3-mixed" else 3e-3), (name, (num / den) ** 0.5)
        print(f"{name} {mode}: gradient error over all parameters {(num / den) ** 0.5:.2e}")
    bad = []
    for k, g in grads.items():
        got = eng.ps.g(k).cpu() / S
        if k.endswith("in_proj_bias"):     # the K-bias third is zero in exact arithmetic (softmax shift invariance)
            d = got.numel() // 3
            got, g = torch.cat([got[:d], got[2 * d:]]), torch.cat([g[:d], g[2 * d:]])
        e = float((got - g).norm())
        if e > gtol * float(g.norm()) + 1e-5 * gtol * gmax:
            bad.append((k, e / (float(g.norm()) + 1e-30)))
    assert not bad, (name, mode, bad[:8], len(bad))
    print(f"{name} {mode}: logits rel err {err:.2e}, ids equal {bool(torch.equal(ids, rid))}, "
          f"undecidable positions {int((margin <= 2 * err * scale).sum())} of {margin.numel()}")
    from tests.conftest import record_parity
    record_parity("test_shape_parity_forward_backward_vs_oracle", workload=name, mode=mode, batch=2, weights="fresh init",
                  logits_rel_err=err, positions=int(margin.numel()), ids_differ=int((ids != rid).sum()),
                  undecidable=int((margin <= 2 * err * scale).sum()))


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x3-mixed", "bf16"])
def test_dropout_keep_bits_equal_rehash(mode):
    """Training step with dropout 0.1 at the c2 shape: the backward kernels reading the forward's keep-bit tensor give the
    logits, the loss and (to the rounding of the gradient storage format) the gradients of the kernels that re-hash every
    score: same dropout stream, so the two are the same function."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodalanalytical_amd import ops
    from multimodalanalytical_amd.engine import Seq2SeqEngine
    wl, cfg, inputs, sd, _ref, _ = _case("c2")    # _ : the oracle's gradient dict = the names of the trained parameters
    cfg = dict(cfg, dropout=0.1)
    enc, am, dec, dm, labels = inputs
    res = []
    for keep_bits in (True, False):
        eng = Seq2SeqEngine(cfg, wl["data"], "Smiles", wl["data"]["Smiles"]["vocab_size"], device=DEV,
                            compute_dtype=_dtype(mode), seed=5, backward_dtype=_bdtype(mode))
        eng.keep_bits = keep_bits
        eng.attn_bwd_flags = 16384 | 32768   # both paths on the 32 x 32 x 16 backward kernels (round 5: the defaults differ by dropout path)
        eng.xattn_fused = False              # (round 6: the fused cross-attention backward exists for the keep-bit path only and keeps dO unrounded --
                                             # another rounding of dQ, 4 % of decoder.layers.0.norm2.weight's small bf16 gradient: not "the same kernels")
        eng.load_state_dict(sd)
        eng.train()
        out = eng.forward(_to(enc), am.to(DEV), dec.to(DEV), dm.to(DEV), labels.to(DEV), backward=True)
        res.append((float(out["loss"]), {k: eng.ps.g(k).cpu().clone() for k in _.keys()}, out["logits"].float().cpu().clone()))
    (l1, g1, y1), (l0, g0, y0) = res
    assert torch.equal(y1, y0)                      # the forward kernels differ only in what they store
    assert abs(l1 - l0) <= 1e-6 * abs(l0), (l1, l0)  # (the mean over tokens is an atomic sum: last-bit run-to-run noise)
    # gradients: same function, same keep decisions (dK / dV are bit-equal kernel by kernel, tools/experiments/check_bits.py);
    # the dQ kernels round `keep * scale * dP - delta` in a different order (fma contraction), which flips last bits of the
    # stored gradients: 1 ulp of the storage format = 2e-5-level in pair mode, bf16-level (4e-3) in single-pass mode
    tol = 2e-5 if mode == "bf16x3" else 1e-2       # (mixed: bf16 backward kernels, the forward's bits)
    for k in g1:
        d, n = float((g1[k] - g0[k]).norm()), float(g0[k].norm())
        assert d <= tol * n + 1e-12, (k, d, n)


def test_labels_on_padded_decoder_rows_keep_their_gradient():
    """The backward leaves padded positions out (exact zeros) -- but whether a padded DECODER row is dead depends on the labels, which
    are the caller's: a batch whose labels are set on padded decoder rows must still give the oracle's gradients (c3, fp16)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodalanalytical_amd.engine import Seq2SeqEngine
    wl, cfg, inputs, sd, _, _ = _case("c3")
    enc, am, dec, dm, labels = inputs
    labels = labels.clone()
    assert bool((dm == 0).any()) and bool(((labels == -100) & (dm == 0)).any())
    labels[dm == 0] = 7                                   # every padded decoder row now carries a loss term
    leaf = {k: v.clone().requires_grad_(not k.endswith("pos_enc")) for k, v in sd.items()}
    ref = O.model_forward(leaf, cfg, wl["data"], "Smiles", enc, am, dec, dm, labels)
    ref["loss"].backward()
    eng = Seq2SeqEngine(cfg, wl["data"], "Smiles", wl["data"]["Smiles"]["vocab_size"], device=DEV, compute_dtype=torch.float16, seed=5)
    eng.load_state_dict(sd)
    out = eng.forward(_to(enc), am.to(DEV), dec.to(DEV), dm.to(DEV), labels.to(DEV), backward=True)
    torch.testing.assert_close(out["loss"].cpu(), ref["loss"].detach(), rtol=2e-3, atol=2e-3)
    S = float(eng.scaler[0])
    num = den = 0.0
    for k, v in leaf.items():
        if v.grad is None or k.endswith("in_proj_bias"):
            continue
        got, want = eng.ps.g(k).cpu() / S, v.grad
        if k == "embedding.embedding_layer_dict.Smiles.weight":
            # nn.Embedding(padding_idx=pad) never gives the pad row a gradient (modeling/utils.py:102-106); the oracle's plain
            # gather does, which only shows in this artificial batch: compare the other rows, and hold the engine to torch's rule
            assert float(got[0].abs().max()) == 0.0
            got, want = got[1:], want[1:]
        num += float((got - want).norm()) ** 2; den += float(want.norm()) ** 2
    assert (num / den) ** 0.5 < 3e-3, (num / den) ** 0.5
