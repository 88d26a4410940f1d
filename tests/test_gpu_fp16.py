"""GPU: the fp16 precision mode (AFM_F16 operands, one v_mfma_*_f16 pass per product, fp32 accumulate, dynamic loss scaling):
the reference's own GPU arithmetic (trainer/trainer.py:69, Lightning "16-mixed" = fp16 autocast + GradScaler).  Every kernel
family that takes the dtype against an fp64 reference on fp16-rounded inputs, the loss scaler against torch.amp.GradScaler's
rules, and the training loop against the reference goldens (parameters after two optimiser steps)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import afm_oracle as O  # noqa: E402
from tests import golden_io as G  # noqa: E402
from tests.dropmask import keep_mask, keep_mask16  # noqa: E402
from tests.test_gpu_ops import _attn_case, _attn_ref, close, dev, rnd  # noqa: E402

DEV = "cuda:0"
H16 = torch.float16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodalanalytical_amd import ops as _ops
    return _ops


# ------------------------------------------------------------------ GEMMs
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (100, 200, 72), (1000, 1536, 512), (129, 24, 64), (4096, 2048, 512), (70000, 384, 128)])
@pytest.mark.parametrize("cdt", [H16, torch.float32])
def test_gemm_f16_nt(ops, M, N, K, cdt):
    a, w, bias = rnd(M, K, seed=1).half(), rnd(N, K, seed=2).half(), rnd(N, seed=3)
    c = torch.empty(M, N, dtype=cdt, device=DEV)
    ops.gemm(dev(a), dev(w), c, bias=dev(bias), algo=2)
    assert ops.last_algo() == "mfma_nt"
    ref = a.double() @ w.double().T + bias.double()
    tol = dict(rtol=2e-3, atol=2e-3 * math.sqrt(K) / 4) if cdt == H16 else dict(rtol=1e-4, atol=2e-4 * math.sqrt(K) / 8)
    close(c, ref, **tol)


def test_gemm_f16_nt_identity_asymmetric(ops):
    n = 128
    a = torch.eye(n).half()
    w = (torch.arange(n * n).view(n, n) % 251 - 125).float().half()
    c = torch.empty(n, n, device=DEV)
    ops.gemm(dev(a), dev(w), c, algo=2)
    close(c, w.float().T, 0, 0)


@pytest.mark.parametrize("variant", [12, 13, 22, 24, 25, 28, 100])
def test_gemm_f16_nt_variants(ops, variant):
    M, N, K = 1024, 1024, 256
    a, w, bias = rnd(M, K, seed=1).half(), rnd(N, K, seed=2).half(), rnd(N, seed=3)
    c = torch.empty(M, N, dtype=H16 if variant == 28 else torch.float32, device=DEV)
    ops.gemm(dev(a), dev(w), c, bias=dev(bias), algo=2, variant=variant)
    tol = dict(rtol=2e-3, atol=1e-2) if variant == 28 else dict(rtol=1e-4, atol=2e-4 * math.sqrt(K) / 8)
    close(c, a.double() @ w.double().T + bias.double(), **tol)


@pytest.mark.parametrize("M,N,K,bias", [(256, 256, 128, True), (512, 768, 192, True), (256 * 37, 256 * 3, 320, True), (4096, 1024, 512, False),
                                        (16384, 512, 2048, True), (16384, 1536, 512, True)])
def test_gemm_f16_nt_pingpong(ops, M, N, K, bias):
    """The ping-pong 256 x 256 kernel (variant 30, csrc/afm_gemm_pp_impl.h): stream starts, tile boundaries, tails shorter than the
    ring, uneven tile counts per workgroup -- against fp64, and bit-equal to the loader-wave kernel when no bias is added (with a bias
    the accumulators START from it there: same sum, other rounding order)."""
    a, w = rnd(M, K, seed=1).half(), (rnd(N, K, seed=2) * 0.1).half()
    b = rnd(N, seed=3) if bias else None
    c = torch.full((M, N), float("nan"), dtype=H16, device=DEV)
    ops.gemm(dev(a), dev(w), c, bias=None if b is None else dev(b), variant=30)
    assert ops.last_algo() == "mfma_nt_pp"
    ref = a.double() @ w.double().T + (b.double() if bias else 0.0)
    close(c, ref, rtol=2e-3, atol=2e-3 * math.sqrt(K) / 4)
    c2 = torch.empty_like(c)
    ops.gemm(dev(a), dev(w), c2, bias=None if b is None else dev(b), variant=24)
    if not bias:
        assert torch.equal(c, c2)
    c3 = torch.full_like(c, float("nan"))
    ops.gemm(dev(a), dev(w), c3, bias=None if b is None else dev(b), variant=30)       # same bits run to run (no race on the ring)
    assert torch.equal(c, c3)


@pytest.mark.parametrize("M,N,K,bias,dt", [(256, 256, 256, True, H16), (512, 768, 384, True, H16), (256 * 37, 256 * 3, 640, True, H16),
                                           (4096, 1024, 512, False, H16), (16384, 512, 2048, True, H16), (8192, 768, 3072, True, torch.bfloat16)])
def test_gemm_nt_four_wave_kernel_is_bit_identical_to_the_pingpong_kernel(ops, M, N, K, bias, dt):
    """The four-wave 256 x 256 kernel (variant 40, csrc/afm_gemm_w4_impl.h: 128 x 128 per wave, fragment prefetch and LDS-DMA between
    the wave's own MFMAs, tied inline-asm MFMAs on a[0:255]): stream starts, tile boundaries, workgroups without a tile, tails past the
    stream's end -- against fp64, bit-equal to the ping-pong kernel (same products, same accumulation order), same bits run to run."""
    a, w = rnd(M, K, seed=1).to(dt), (rnd(N, K, seed=2) * 0.1).to(dt)
    b = rnd(N, seed=3) if bias else None
    c = torch.full((M, N), float("nan"), dtype=dt, device=DEV)
    ops.gemm(dev(a), dev(w), c, bias=None if b is None else dev(b), variant=40)
    assert ops.last_algo() == "mfma_nt_w4"
    ref = a.double() @ w.double().T + (b.double() if bias else 0.0)
    tol = 2e-3 if dt == H16 else 1.6e-2
    close(c, ref, rtol=tol, atol=tol * math.sqrt(K) / 4)
    c2 = torch.empty_like(c)
    ops.gemm(dev(a), dev(w), c2, bias=None if b is None else dev(b), variant=30)
    assert ops.last_algo() == "mfma_nt_pp" and torch.equal(c, c2)
    c3 = torch.full_like(c, float("nan"))
    ops.gemm(dev(a), dev(w), c3, bias=None if b is None else dev(b), variant=40)
    assert torch.equal(c, c3)


@pytest.mark.parametrize("K,algo", [(512, "mfma_nt_pp"), (1536, "mfma_nt_w4")])
def test_gemm_f16_nt_pingpong_is_selected_and_takes_the_row_hint(ops, K, algo):
    """Automatic dispatch at step-sized shapes (K = 512 with a wide N: the ping-pong kernel; K >= 768: the four-wave kernel), and
    afm_gemm_desc.k_live through the kernels' tile lists (dead tiles written as zeros)."""
    M, N = 65536, (1024 if K == 512 else 512)
    live = torch.ones(M // 64, dtype=torch.uint8)
    live[8:24] = 0; live[29] = 0; live[400:700] = 0; live[1000:] = 0
    rl = live.repeat_interleave(64).bool()
    a = rnd(M, K, seed=1) * 0.5; a[~rl] = 0.0
    ad, wd = dev(a, H16), dev(rnd(N, K, seed=2) * 0.1, H16)
    outs = []
    for hint in (None, dev(live)):
        c = torch.full((M, N), 3.0, dtype=H16, device=DEV)
        ops.gemm(ad, wd, c, k_live=hint)
        assert ops.last_algo() == algo
        outs.append(c)
    assert torch.equal(outs[0], outs[1])
    assert float(outs[1][~rl.to(DEV)].abs().max()) == 0.0
    ref = torch.empty_like(outs[0]); ops.gemm(ad, wd, ref, variant=24)
    assert torch.equal(outs[0], ref)


@pytest.mark.parametrize("M,N", [(1024, 512), (4096, 2048)])     # 256x128 loader-wave form / 256x256 form (auto-selected)
def test_gemm_f16_gelu_save_grad_pair(ops, M, N):
    """FFN up-projection forward with the stored backward factor (act 4) and the dgrad that multiplies by it (act 5)."""
    K = 128
    a, w, bias = rnd(M, K, seed=1).half(), rnd(N, K, seed=2).half(), rnd(N, seed=3)
    t = a.double() @ w.double().T
    p, seed, site = 0.1, 31, 9
    keep = torch.from_numpy(keep_mask(p, seed, site, M * N)).view(M, N)
    c, gp = torch.zeros(M, N, dtype=H16, device=DEV), torch.zeros(M, N, dtype=H16, device=DEV)
    ops.gemm(dev(a), dev(w), c, bias=dev(bias), act=4, pre_act=gp, dropout=ops.drop(p, seed, site), algo=2)
    tr = (t + bias.double()).requires_grad_(True)
    O.gelu(tr).backward(torch.ones(M, N, dtype=torch.float64))
    close(c, O.gelu(tr.detach()) * keep / (1 - p), rtol=3e-3, atol=4e-3)
    close(gp, tr.grad * keep / (1 - p), rtol=3e-3, atol=4e-3)
    c2 = torch.zeros_like(c)
    ops.gemm(dev(a), dev(w), c2, act=5, pre_act=gp, algo=2)
    close(c2, t * gp.float().cpu().double(), rtol=3e-3, atol=1e-2)


def test_gemm_f16_glu_fused(ops):
    """Gated FFN through the fused epilogues (act 6 / 7 / 8) on fp16 operands, against the unfused formulas."""
    M, f, d = 512, 256, 128
    h = rnd(M, d, seed=1).half()
    w1, wg = rnd(f, d, seed=2) * 0.2, rnd(f, d, seed=3) * 0.2
    bias = rnd(2 * f, seed=4) * 0.1
    wcat = torch.cat([w1, wg], 0)
    w_il, wt_il = torch.empty(2 * f, d, dtype=H16, device=DEV), torch.empty(d, 2 * f, dtype=H16, device=DEV)
    ops.cast_weights(dev(wcat), w_il, wt_il, glu_rows=f)
    g = torch.empty(M, f, dtype=H16, device=DEV); uv = torch.empty(M, 2 * f, dtype=H16, device=DEV)
    ops.gemm(dev(h), w_il, g, bias=dev(bias), act=7, pre_act=uv, glu_rows=f, algo=2)
    assert "glu" in ops.last_algo()
    u = h.double() @ w1.half().double().T + bias[:f].double()
    v = h.double() @ wg.half().double().T + bias[f:].double()
    close(g, O.gelu(u) * v, rtol=3e-3, atol=4e-3)


def test_gemm_f16_glu_fused_tile_forms_agree(ops):
    """The gated-FFN epilogues on the 256 x 256 persistent kernel (the default for the forward forms over >= 1 024 such tiles;
    afm_gemm_desc.reserved = 28 forces it) write what the 256 x 128 loader-wave kernel (reserved = 24) writes, bit for bit: up-projection
    with GELU pair + dropout + the two stored factors, the same without stored factors, and the data gradient x stored."""
    M, f, d = 16384, 2048, 256          # 64 x 16 = 1 024 tiles of 256 x 256 over the 2f-wide accumulators
    h = dev(rnd(M, d, seed=1), H16)
    w_il, wt_il = torch.empty(2 * f, d, dtype=H16, device=DEV), torch.empty(d, 2 * f, dtype=H16, device=DEV)
    ops.cast_weights(dev(rnd(2 * f, d, seed=2) * 0.1), w_il, wt_il, glu_rows=f)
    bias = dev(rnd(2 * f, seed=4) * 0.1)
    w2t = dev(rnd(f, d, seed=5) * 0.1, H16)
    dy = dev(rnd(M, d, seed=6) * 0.01, H16)
    dr = ops.drop(0.1, 7, 3)
    out = {}
    for v in (24, 28, 0):
        g = torch.full((M, f), float("nan"), dtype=H16, device=DEV); uv = torch.full((M, 2 * f), float("nan"), dtype=H16, device=DEV)
        ops.gemm(h, w_il, g, bias=bias, act=7, pre_act=uv, glu_rows=f, dropout=dr, variant=v)
        assert "glu" in ops.last_algo()
        g2 = torch.full_like(g, float("nan"))
        ops.gemm(h, w_il, g2, bias=bias, act=6, glu_rows=f, dropout=dr, variant=v)
        duv = torch.full((M, 2 * f), float("nan"), dtype=H16, device=DEV)
        ops.gemm(dy, w2t, duv, act=8, pre_act=uv, glu_rows=f, variant=v)
        out[v] = (g, uv, g2, duv)
    for v in (28, 0):
        for a, b in zip(out[24], out[v]):
            assert bool(torch.isfinite(b.float()).all()) and torch.equal(a, b), v


@pytest.mark.parametrize("R,M,N", [(512, 128, 128), (4096, 1536, 512), (777, 24, 64), (16384, 520, 200), (131072, 512, 256)])
def test_gemm_f16_tn_wgrad(ops, R, M, N):
    dy, x = (rnd(R, M, seed=1) * 0.5).half(), rnd(R, N, seed=2).half()
    g0 = rnd(M, N, seed=3)
    g = dev(g0).clone()
    b0 = rnd(M, seed=4); gb = dev(b0).clone()
    ops.gemm(dev(dy), dev(x), g, trans_a=True, trans_b=False, accumulate=True, algo=2, a_colsum=gb)
    assert ops.last_algo().startswith("mfma_tn")
    close(g, g0.double() + dy.double().T @ x.double(), 1e-4, 2e-4 * math.sqrt(R) / 4)
    close(gb, b0.double() + dy.double().sum(0), 1e-4, 2e-4 * math.sqrt(R) / 4, "fused bias gradient")


@pytest.mark.parametrize("dt", [H16, torch.bfloat16])
@pytest.mark.parametrize("R", [4096, 16384 + 64])
def test_gemm_group_wgrads_share_one_launch(ops, dt, R):
    """afm_gemm_group: a layer's weight gradients in one grid (one split-K budget) equal the one-by-one launches; problems it does
    not fuse (small, NT, fp32) run through afm_gemm in order; more than 8 eligible problems take a second launch."""
    shapes = [(512, 512), (1536, 512), (512, 2048), (264, 520), (2048, 512), (256, 256), (512, 264), (768, 256), (512, 512), (256, 1024)]
    probs, keep = [], []
    for i, (M, N) in enumerate(shapes):
        dy, x = (rnd(R, M, seed=10 + i) * 0.5).to(dt), rnd(R, N, seed=40 + i).to(dt)
        g0, b0 = rnd(M, N, seed=70 + i), rnd(M, seed=90 + i)
        glu = M // 2 if i == 4 else 0              # one gated problem: rows de-interleaved into the [W1 ; Wg] order
        probs.append((dy, x, g0, b0 if i % 3 else None, glu))
    # not fused: a small weight gradient, an NT product, an fp32 weight gradient
    sm_dy, sm_x, sm_g0 = (rnd(R, 64, seed=5) * 0.5).to(dt), rnd(R, 128, seed=6).to(dt), rnd(64, 128, seed=7)
    nt_a, nt_w = rnd(300, 128, seed=8).to(dt), rnd(200, 128, seed=9).to(dt)
    descs, outs = [], []
    for dy, x, g0, b0, glu in probs[:5]:
        g = dev(g0).clone(); gb = None if b0 is None else dev(b0).clone()
        a, b = dev(dy), dev(x); keep += [a, b]
        descs.append(ops.gemm_desc(a, b, g, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb, glu_rows=glu)); outs.append((g, gb))
    sm_g = dev(sm_g0).clone(); a, b = dev(sm_dy), dev(sm_x); keep += [a, b]
    descs.append(ops.gemm_desc(a, b, sm_g, trans_a=True, trans_b=False, accumulate=True))
    nt_c = torch.empty(300, 200, dtype=dt, device=DEV); a, b = dev(nt_a), dev(nt_w); keep += [a, b]
    descs.append(ops.gemm_desc(a, b, nt_c))
    for dy, x, g0, b0, glu in probs[5:]:
        g = dev(g0).clone(); gb = None if b0 is None else dev(b0).clone()
        a, b = dev(dy), dev(x); keep += [a, b]
        descs.append(ops.gemm_desc(a, b, g, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb, glu_rows=glu)); outs.append((g, gb))
    ops.reset_algo_log()
    ops.gemm_group(descs)
    assert ops.last_algo() == "mfma_tn_group256"
    tol = (1e-4, 2e-4 * math.sqrt(R) / 4)
    for (dy, x, g0, b0, glu), (g, gb) in zip(probs, outs):
        ref = dy.double().T @ x.double()
        refb = dy.double().sum(0)
        if glu:      # row m of the interleaved product (4 rows of W1, 4 gate rows, ...) lands in the [W1 ; Wg] layout (include/afm_hip.h)
            m = torch.arange(ref.shape[0])
            dest = ((m >> 3) << 2) + (m & 3) + ((m >> 2) & 1) * glu
            r2, b2 = torch.empty_like(ref), torch.empty_like(refb)
            r2[dest], b2[dest] = ref, refb
            ref, refb = r2, b2
        close(g, g0.double() + ref, *tol)
        if gb is not None:
            close(gb, b0.double() + refb, *tol, "fused bias gradient")
    close(sm_g, sm_g0.double() + sm_dy.double().T @ sm_x.double(), *tol)
    close(nt_c, nt_a.double() @ nt_w.double().T, 1e-2 if dt == torch.bfloat16 else 2e-3, 0.2 if dt == torch.bfloat16 else 0.03)


def test_gemm_group_four_wave_unit_at_the_step_size(ops):
    """The c2 encoder layer's four weight gradients (131 072 tokens) through the four-wave unit (108: built and measured in round 5,
    not the default) and the eight-wave one: each against fp64, bias gradients included, and one gated problem with the de-interleave."""
    R = 131072
    gen = torch.Generator(device=DEV).manual_seed(5)
    shapes = [(512, 512, 0), (1536, 512, 0), (512, 2048, 0), (2048, 512, 1024)]
    ten = [((torch.randn(R, M, device=DEV, generator=gen) * 0.05).half(), torch.randn(R, N, device=DEV, generator=gen).half(), glu) for M, N, glu in shapes]
    for form, algo in ((108, "mfma_tn_groupw4"), (0, "mfma_tn_group256")):
        outs = [(torch.zeros(dy.shape[1], x.shape[1], device=DEV), torch.zeros(dy.shape[1], device=DEV)) for dy, x, _ in ten]
        ops.gemm_group([ops.gemm_desc(dy, x, g, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb, glu_rows=glu, variant=form)
                        for (dy, x, glu), (g, gb) in zip(ten, outs)])
        assert ops.last_algo() == algo
        for (dy, x, glu), (g, gb) in zip(ten, outs):
            ref, refb = (dy.double().T @ x.double()).cpu(), dy.double().sum(0).cpu()
            if glu:
                m = torch.arange(ref.shape[0])
                dest = ((m >> 3) << 2) + (m & 3) + ((m >> 2) & 1) * glu
                r2, b2 = torch.empty_like(ref), torch.empty_like(refb)
                r2[dest], b2[dest] = ref, refb
                ref, refb = r2, b2
            close(g, ref, 1e-4, 2e-4 * math.sqrt(R) / 4)
            close(gb, refb, 1e-4, 2e-4 * math.sqrt(R) / 4, "fused bias gradient")


def test_gemm_group_matches_single_launches_at_the_step_size(ops):
    """The c2 encoder layer's four weight gradients (131 072 tokens): grouped == one by one (fp32 atomics in another order)."""
    R = 131072
    shapes = [(512, 512), (1536, 512), (512, 2048), (2048, 512)]
    gen = torch.Generator(device=DEV).manual_seed(3)
    descs, outs, singles = [], [], []
    for M, N in shapes:
        dy = (torch.randn(R, M, device=DEV, generator=gen) * 0.05).half(); x = torch.randn(R, N, device=DEV, generator=gen).half()
        g, gb = torch.zeros(M, N, device=DEV), torch.zeros(M, device=DEV)
        g1, gb1 = torch.zeros(M, N, device=DEV), torch.zeros(M, device=DEV)
        ops.gemm(dy, x, g1, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb1)
        descs.append(ops.gemm_desc(dy, x, g, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb))
        outs.append((g, gb, dy, x)); singles.append((g1, gb1))
    ops.gemm_group(descs)
    assert ops.last_algo() == "mfma_tn_group256"
    for (g, gb, dy, x), (g1, gb1) in zip(outs, singles):
        torch.testing.assert_close(g, g1, rtol=1e-4, atol=2e-3)
        torch.testing.assert_close(gb, gb1, rtol=1e-4, atol=2e-3)
    g, _, dy, x = outs[0]
    close(g, dy.double().T.cpu() @ x.double().cpu(), 1e-4, 2e-4 * math.sqrt(R) / 4)


def test_gemm_f16_generic_odd_shapes(ops):
    """Shapes outside the MFMA kernels (the SMILES vocabulary, patch sizes) run on the exact-fp32 FMA kernel with fp16 I/O."""
    M, N, K = 37, 26, 75
    a, w = rnd(M, K, seed=1).half(), rnd(N, K, seed=2).half()
    c = torch.empty(M, N, dtype=H16, device=DEV)
    ops.gemm(dev(a), dev(w), c)
    assert ops.last_algo().startswith("generic")
    close(c, a.double() @ w.double().T, 2e-3, 2e-3 * math.sqrt(K))


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,H,Tq,Tk,causal,pad,pdrop", [
    (2, 2, 128, 128, True, True, 0.0), (2, 3, 100, 200, False, True, 0.0), (1, 2, 192, 192, True, False, 0.0),
    (2, 2, 128, 256, False, True, 0.1), (1, 2, 160, 160, True, True, 0.1), (2, 2, 130, 520, False, True, 0.1), (2, 2, 256, 56, False, True, 0.1),
    (1, 2, 320, 320, True, True, 0.1), (2, 2, 512, 1024, False, "blocks", 0.1), (1, 1, 1024, 1024, False, False, 0.0), (1, 2, 300, 300, False, False, 0.0)])
def test_attention_f16(ops, B, H, Tq, Tk, causal, pad, pdrop):
    dh, dt = 64, H16
    q, k, v, key_pad = _attn_case(B, H, Tq, Tk, dh, causal, bool(pad), seed=10)
    if pad == "blocks":   # whole 64-key tiles of padding, as in multimodal batches
        key_pad = torch.zeros(B, Tk, dtype=torch.bool)
        key_pad[0, 64:192] = True; key_pad[0, 300:] = True
        key_pad[1, 128:160] = True; key_pad[1, 448:512] = True
    if Tq == 1024:        # a score spike far above the running maximum in a LATE tile: the deferred-rescale branch must fire
        k[0, 700, 0] = q[0, 5, 0] * 6.0
        k[0, 900, 0] = q[0, 37, 0] * 9.0
    q, k, v = q.half().float(), k.half().float(), v.half().float()
    D = H * dh
    qd, kd, vd = (dev(t.reshape(-1, D), dt) for t in (q, k, v))
    kp = None if key_pad is None else dev(key_pad.to(torch.uint8))
    seed, site = 4242, 3
    keep, dscale = None, 1.0
    if pdrop > 0:
        km, dscale = keep_mask16(pdrop, seed, site, B * H * Tq * Tk, Tk)
        keep = torch.from_numpy(km).view(B, H, Tq, Tk)
    res = {}
    for algo in (1, 2):
        o = torch.empty(B * Tq, D, dtype=dt, device=DEV); lse = torch.empty(B * H * Tq, device=DEV)
        shp = ops.attn_shape(B, H, Tq, Tk, dh, dt, D, D, D, D, kp, causal, ops.drop(pdrop, seed, site), algo=algo)
        if Tq in (256, 320, 512, 1024):      # these cases run the 8-wave staggered forward / dQ kernels (opt-in forms), Tq = 300 the default ones
            shp.reserved = 16
        if pdrop > 0 and algo == 2 and Tq % 2 == 0:
            ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=DEV))
        ops.attn_fwd(shp, qd, kd, vd, o, lse)
        assert ops.last_algo() == ("attn_generic" if algo == 1 else "attn_mfma")
        res[algo] = (o, lse, shp)
    qr, kr, vr, ref = _attn_ref(q, k, v, key_pad, causal, keep, dscale)
    ref_o = ref.transpose(1, 2).reshape(B * Tq, D)
    close(res[2][0], ref_o, 3e-3, 3e-3, "mfma fwd vs oracle")
    close(res[1][0], ref_o, 3e-3, 3e-3, "generic fwd vs oracle")
    close(res[2][1], res[1][1].cpu(), 1e-3, 1e-3, "lse mfma vs generic")
    do = rnd(B * Tq, D, seed=9).half().float()
    ref.backward(do.double().view(B, Tq, H, dh).transpose(1, 2))
    o, lse, shp = res[2]
    dq, dk, dv = (torch.empty(n, D, dtype=dt, device=DEV) for n in (B * Tq, B * Tk, B * Tk))
    ops.attn_bwd(shp, qd, kd, vd, o, dev(do, dt), lse, torch.empty_like(lse), dq, dk, dv, D, D, D)
    assert ops.last_algo() == "attn_mfma"
    for name, got, r, T in (("dq", dq, qr, Tq), ("dk", dk, kr, Tk), ("dv", dv, vr, Tk)):
        want = r.grad.transpose(1, 2).reshape(B * T, D)
        err = float((got.float().cpu().double() - want).norm() / want.norm())
        assert err < 4e-3, (name, err)


@pytest.mark.parametrize("dt", [H16, torch.bfloat16])
@pytest.mark.parametrize("B,H,Tq,Tk,causal,pad", [(2, 2, 128, 256, False, True), (1, 2, 160, 160, True, True), (2, 2, 130, 520, False, True),
                                                  (2, 2, 256, 56, False, False), (2, 8, 1024, 1024, False, False)])
def test_attention_keep_bits_filled_ahead(ops, dt, B, H, Tq, Tk, causal, pad):
    """afm_attn_drop_bits_fill + a forward that READS the keep-bit tensor == the forward that hashes and writes it, bit for bit
    (output, lse, and the bits of every score inside the tensors); the backward takes either tensor."""
    dh, D = 64, H * 64
    q, k, v, key_pad = _attn_case(B, H, Tq, Tk, dh, causal, pad, seed=21)
    qd, kd, vd = (dev(t.reshape(-1, D), dt) for t in (q, k, v))
    kp = None if key_pad is None else dev(key_pad.to(torch.uint8))
    dr = ops.drop(0.1, 777, 5)
    nw = ops.attn_drop_bits_words(B, H, Tq, Tk)
    outs = []
    for ahead in (False, True):
        o = torch.empty(B * Tq, D, dtype=dt, device=DEV); lse = torch.empty(B * H * Tq, device=DEV)
        bits = torch.zeros(nw, dtype=torch.int64, device=DEV)
        shp = ops.attn_shape(B, H, Tq, Tk, dh, dt, D, D, D, D, kp, causal, dr, algo=2)
        ops.attn_set_drop_bits(shp, bits)
        if ahead:
            assert ops.attn_fill_drop_bits(shp) and shp.reserved & 32
        ops.attn_fwd(shp, qd, kd, vd, o, lse)
        assert ops.last_algo() == "attn_mfma"
        outs.append((o, lse, bits, shp))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # bits: 16 lane masks per (batch*head, 32-query block, 32-key block); compare the scores that exist (q < Tq, key < Tk), in the
    # tiles the forward visits (it skips tiles above the causal diagonal and tiles of nothing but padding)
    nq32, nk32 = ((Tq + 127) // 128) * 4, ((Tk + 63) // 64) * 2
    b0 = outs[0][2].view(B * H, nq32, nk32, 16).cpu().numpy().view("uint64")
    b1 = outs[1][2].view(B * H, nq32, nk32, 16).cpu().numpy().view("uint64")
    import numpy as np
    lane = np.arange(64)
    checked = 0
    for qb in range(nq32):
        for kb in range(nk32):
            if causal and (kb // 2) * 64 > qb * 32 + 31:      # the wave's first query block row ends before the tile's first key
                continue
            for r in range(16):
                key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                qq = qb * 32 + (lane & 31)
                valid = (qq < Tq) & (key < Tk)
                m = np.uint64(sum(1 << int(i) for i in lane[valid]))
                assert np.array_equal(b0[:, qb, kb, r] & m, b1[:, qb, kb, r] & m), (qb, kb, r)
                checked += 1
    assert checked > 0      # (no case here pads a whole 64-key tile: the emitting forward would have skipped it)
    # the backward is indifferent to who wrote the tensor
    do = dev(rnd(B * Tq, D, seed=9), dt)
    grads = []
    for o, lse, bits, shp in outs:
        dq, dk, dv = (torch.empty(n, D, dtype=dt, device=DEV) for n in (B * Tq, B * Tk, B * Tk))
        delta = torch.empty_like(lse)
        ops.attn_bwd(shp, qd, kd, vd, o, do, lse, delta, dq, dk, dv, D, D, D)
        grads.append((dq, dk, dv))
    for a, b in zip(*grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dt", [H16, torch.bfloat16])
@pytest.mark.parametrize("B,H,Tq,Tk,causal,pad,pdrop", [(2, 2, 128, 256, False, True, 0.1), (1, 2, 160, 160, True, True, 0.1), (2, 2, 130, 520, False, True, 0.0),
                                                        (2, 2, 256, 56, False, False, 0.1), (1, 8, 1024, 1024, False, False, 0.1)])
def test_attention_forward_16x16x32_form(ops, dt, B, H, Tq, Tk, causal, pad, pdrop):
    """The forward restated on v_mfma_f32_16x16x32 (csrc/afm_attn_fwd16_impl.h; afm_attn_shape.reserved & 1024, & 2048 its three-workgroup
    build: A / B forms, measured slower with dropout and left off) against the shipped forward: O and lse to rounding (another
    accumulation order), the keep-bit tensor it writes bit for bit, and reading a tensor filled ahead gives the same bits of O."""
    dh, D = 64, H * 64
    q, k, v, key_pad = _attn_case(B, H, Tq, Tk, dh, causal, pad, seed=23)
    qd, kd, vd = (dev(t.reshape(-1, D), dt) for t in (q, k, v))
    kp = None if key_pad is None else dev(key_pad.to(torch.uint8))
    dr = ops.drop(pdrop, 778, 5) if pdrop > 0 else ops.NO_DROP
    nw = ops.attn_drop_bits_words(B, H, Tq, Tk)
    res = {}
    for name, flag in (("ref", 0), ("m16", 1024), ("m16occ3", 1024 | 2048)):
        o = torch.full((B * Tq, D), float("nan"), dtype=dt, device=DEV); lse = torch.full((B * H * Tq,), float("nan"), device=DEV)
        bits = torch.zeros(nw, dtype=torch.int64, device=DEV)
        shp = ops.attn_shape(B, H, Tq, Tk, dh, dt, D, D, D, D, kp, causal, dr, algo=2)
        if pdrop > 0:
            ops.attn_set_drop_bits(shp, bits)
        shp.reserved |= flag
        ops.attn_fwd(shp, qd, kd, vd, o, lse)
        assert ops.last_algo() == "attn_mfma"
        res[name] = (o, lse, bits.clone())
        if pdrop > 0 and flag:
            o2 = torch.full_like(o, float("nan")); lse2 = torch.full_like(lse, float("nan"))
            assert ops.attn_fill_drop_bits(shp) and shp.reserved & 32
            ops.attn_fwd(shp, qd, kd, vd, o2, lse2)
            assert torch.equal(o, o2) and torch.equal(lse, lse2), name
    o0, l0, b0 = res["ref"]
    tol = 4e-3 if dt == H16 else 3e-2
    for name in ("m16", "m16occ3"):
        o1, l1, b1 = res[name]
        assert bool(torch.isfinite(o1.float()).all())
        assert float((o0.float() - o1.float()).abs().max()) <= tol * float(o0.float().abs().max()), name
        inf = torch.isinf(l0)
        assert torch.equal(inf, torch.isinf(l1)) and float((l0[~inf] - l1[~inf]).abs().max()) < 2e-5, name
        assert torch.equal(b0, b1), name


@pytest.mark.parametrize("dt", [H16, torch.bfloat16])
@pytest.mark.parametrize("B,H,Tq,Tk,pad,pdrop", [(2, 4, 128, 1024, False, 0.1), (3, 2, 128, 1000, True, 0.1), (2, 2, 100, 520, True, 0.1),
                                                 (2, 4, 64, 256, False, 0.0), (2, 2, 192, 768, True, 0.1), (5, 8, 40, 2048, True, 0.1)])
def test_attention_dkv_short_query_form(ops, dt, B, H, Tq, Tk, pad, pdrop):
    """The short-query dK / dV kernel (csrc/afm_attn_sq_impl.h, afm_attn_shape.reserved & 65536: every query tile of a head resident in
    LDS, the workgroup walks the key blocks; an A / B form for the decoder's cross-attention shapes) against the general kernels:
    dK / dV to rounding, zero rows at padded keys."""
    dh, D = 64, H * 64
    q, k, v, _ = _attn_case(B, H, Tq, Tk, dh, False, False, seed=41)
    key_pad = torch.zeros(B, Tk, dtype=torch.bool)
    if pad:
        key_pad[0, Tk // 3:] = True
        key_pad[B - 1, Tk - 70:] = True
    qd, kd, vd = (dev(t.reshape(-1, D), dt) for t in (q, k, v))
    kp = dev(key_pad.to(torch.uint8)) if pad else None
    dr = ops.drop(pdrop, 6, 2) if pdrop > 0 else ops.NO_DROP
    dod = dev(rnd(B * Tq, D, seed=12), dt)
    o = torch.empty(B * Tq, D, dtype=dt, device=DEV); lse = torch.empty(B * H * Tq, device=DEV)
    shp = ops.attn_shape(B, H, Tq, Tk, dh, dt, D, D, D, D, kp, False, dr, algo=2)
    if pdrop > 0:
        ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=DEV))
    ops.attn_fwd(shp, qd, kd, vd, o, lse)
    res = []
    for flag in (0, 65536):
        shp.reserved = flag
        dq = torch.empty(B * Tq, D, dtype=dt, device=DEV)
        dk, dv = (torch.full((B * Tk, D), float("nan"), dtype=dt, device=DEV) for _ in range(2))
        ops.attn_bwd(shp, qd, kd, vd, o, dod, lse, torch.empty_like(lse), dq, dk, dv, D, D, D)
        assert ops.last_algo() == "attn_mfma"
        res.append((dk.float(), dv.float()))
    tol = 4e-3 if dt == H16 else 3e-2
    for i in range(2):
        a_, b_ = res[0][i], res[1][i]
        assert bool(torch.isfinite(b_).all())
        assert float((a_ - b_).abs().max()) <= tol * float(a_.abs().max()), i
    if pad:
        dead = dev(key_pad.reshape(-1))
        assert float(res[1][0][dead].abs().max()) == 0.0 and float(res[1][1][dead].abs().max()) == 0.0


@pytest.mark.parametrize("dt", [H16, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
def test_attention_backward_skips_padded_query_rows_exactly(ops, dt, causal):
    """afm_attn_shape.reserved bit 6: self-attention over a padded batch whose padded rows carry zero dO (a training step's
    padded positions).  With the flag the single-pass kernels skip those rows as queries: dQ, dK, dV equal the unflagged run."""
    B, H, T, dh = 3, 2, 512, 64
    D = H * dh
    q, k, v, _ = _attn_case(B, H, T, T, dh, causal, False, seed=31)
    key_pad = torch.zeros(B, T, dtype=torch.bool)
    key_pad[0, 100:] = True; key_pad[1, 333:] = True          # whole 32-query waves / 64-query tiles of padding, and ragged edges
    qd, kd, vd = (dev(t.reshape(-1, D), dt) for t in (q, k, v))
    kp = dev(key_pad.to(torch.uint8))
    dr = ops.drop(0.1, 99, 4)
    do = rnd(B * T, D, seed=9)
    do[key_pad.reshape(-1)] = 0.0                              # what the engine's backward hands over at padded positions
    dod = dev(do, dt)
    res = []
    for flag in (0, 64):
        o = torch.empty(B * T, D, dtype=dt, device=DEV); lse = torch.empty(B * H * T, device=DEV)
        shp = ops.attn_shape(B, H, T, T, dh, dt, D, D, D, D, kp, causal, dr, algo=2)
        ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, T, T), dtype=torch.int64, device=DEV))
        ops.attn_fwd(shp, qd, kd, vd, o, lse)
        shp.reserved |= flag
        dq, dk, dv = (torch.full((B * T, D), 7.0, dtype=dt, device=DEV) for _ in range(3))
        ops.attn_bwd(shp, qd, kd, vd, o, dod, lse, torch.empty_like(lse), dq, dk, dv, D, D, D)
        assert ops.last_algo() == "attn_mfma"
        res.append((dq, dk, dv))
    for a, b in zip(*res):
        assert torch.equal(a, b)                               # (-0 == +0: the skipped rows are written as zeros either way)
    assert float(res[1][0].view(B, T, D)[0, 100:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", [H16, torch.bfloat16])
@pytest.mark.parametrize("B,H,Tq,Tk,pad,pdrop", [(2, 2, 64, 128, False, 0.1), (2, 4, 256, 256, True, 0.1), (3, 2, 192, 320, True, 0.1),
                                                 (2, 4, 512, 512, True, 0.0), (4, 2, 128, 1024, True, 0.1), (1, 8, 1024, 1024, False, 0.1)])
def test_attention_dkv_pipelined_kernel_is_bit_identical(ops, dt, B, H, Tq, Tk, pad, pdrop):
    """csrc/afm_attn_pipe_impl.h (software-pipelined dK/dV kernel: no causal mask, Tq % 64 == 0, keep-bit dropout or none) gives every
    accumulator its products in the round-3 kernel's order: dK and dV equal that kernel's (afm_attn_shape.reserved & 128) bit for bit,
    in the four-wave form (reserved & 16384 since round 5: the default is now its 16 x 16 x 32 restatement, which has another
    accumulation order), with 64 keys per wave (reserved & 512) and with eight waves (reserved & 256), with and without the padded-query
    skip.  The 16 x 16 x 32 forms -- the pipelined default (csrc/afm_attn_pipe16_impl.h) and the round-3 kernel's restatement (reserved &
    4096, csrc/afm_attn_m16_impl.h) -- agree with it to rounding."""
    dh = 64
    D = H * dh
    q, k, v, _ = _attn_case(B, H, Tq, Tk, dh, False, False, seed=41)
    key_pad = torch.zeros(B, Tk, dtype=torch.bool)
    if pad:
        key_pad[0, Tk // 3:] = True
        key_pad[B - 1, Tk - 70:] = True
    qd, kd, vd = dev(q.reshape(-1, D), dt), dev(k.reshape(-1, D), dt), dev(v.reshape(-1, D), dt)
    kp = dev(key_pad.to(torch.uint8)) if pad else None
    dr = ops.drop(pdrop, 5, 2) if pdrop > 0 else ops.NO_DROP
    do = rnd(B * Tq, D, seed=10)
    qskip = pad and Tq == Tk
    if qskip:
        do[key_pad.reshape(-1)] = 0.0
    dod = dev(do, dt)
    o = torch.empty(B * Tq, D, dtype=dt, device=DEV); lse = torch.empty(B * H * Tq, device=DEV)
    shp = ops.attn_shape(B, H, Tq, Tk, dh, dt, D, D, D, D, kp, False, dr, algo=2)
    if pdrop > 0:
        ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=DEV))
    ops.attn_fwd(shp, qd, kd, vd, o, lse)
    res = {}
    for name, flag in (("round3", 128), ("pipe32", 16384), ("pipe64", 512 | 16384), ("pipe8w", 256 | 16384), ("pipe16", 0), ("m16", 4096)):
        shp.reserved = flag | (64 if qskip else 0)
        dq = torch.empty(B * Tq, D, dtype=dt, device=DEV)
        dk, dv = (torch.full((B * Tk, D), float("nan"), dtype=dt, device=DEV) for _ in range(2))
        ops.attn_bwd(shp, qd, kd, vd, o, dod, lse, torch.empty_like(lse), dq, dk, dv, D, D, D)
        assert ops.last_algo() == "attn_mfma"
        res[name] = (dk, dv)
    assert bool(torch.isfinite(res["pipe64"][0].float()).all()) and bool(torch.isfinite(res["pipe64"][1].float()).all())
    for name in ("pipe64", "pipe32", "pipe8w"):
        assert torch.equal(res["round3"][0], res[name][0]) and torch.equal(res["round3"][1], res[name][1]), name
    tol = 4e-3 if dt == H16 else 3e-2
    for name in ("pipe16", "m16"):
        for i in range(2):
            a_, b_ = res["round3"][i].float(), res[name][i].float()
            assert bool(torch.isfinite(b_).all()), name
            assert float((a_ - b_).abs().max()) <= tol * float(a_.abs().max()), (name, i)


@pytest.mark.parametrize("dt", [H16, torch.bfloat16])
def test_cast_weights_batch_equals_the_single_launches(ops, dt):
    """afm_cast_weights_batch (every weight shadow of an optimiser step in one launch) writes what the per-matrix launches write:
    plain copies, transposes, the gated-FFN row interleave, shapes that are not multiples of the 64 x 64 tile."""
    shapes = [(1536, 512, 0), (512, 512, 0), (300, 70, 0), (4096, 512, 2048), (26, 512, 0), (64, 64, 0), (512, 2048, 0)]
    srcs = [dev(rnd(r, c, seed=60 + i)) for i, (r, c, _) in enumerate(shapes)]
    want, entries = [], []
    for i, ((r, c, glu), src) in enumerate(zip(shapes, srcs)):
        d1 = torch.full((r, c), 7.0, dtype=dt, device=DEV) if i % 3 != 1 else None       # (every third item: transpose only)
        t1 = torch.full((c, r), 7.0, dtype=dt, device=DEV)
        if dt == H16 or glu:
            ops.cast_weights(src, d1, t1, glu_rows=glu)
        else:
            ops.cast_bf16(src, d1, t1)
        want.append((d1, t1))
        entries.append((src, None if d1 is None else torch.full_like(d1, 3.0), torch.full_like(t1, 3.0), glu))
    batch = ops.CastBatch(entries, dt)
    batch.run()
    for (d1, t1), (_, d2, t2, _) in zip(want, entries):
        assert torch.equal(t1, t2)
        assert d1 is None or torch.equal(d1, d2)


# ------------------------------------------------------------------ LayerNorm, elementwise, loss
@pytest.mark.parametrize("d", [64, 512, 768])
def test_layernorm_f16(ops, d):
    rows = 300
    x, br = rnd(rows, d, seed=1) * 2 + 0.5, rnd(rows, d, seed=2)
    gam, bet = 1 + 0.1 * rnd(d, seed=3), 0.1 * rnd(d, seed=4)
    brd = dev(br, H16)
    y = torch.empty(rows, d, dtype=H16, device=DEV); xs = torch.empty(rows, d, device=DEV)
    mean = torch.empty(rows, device=DEV); rstd = torch.empty(rows, device=DEV)
    p, seed, site = 0.1, 3, 8
    ops.layernorm_fwd(dev(x), dev(gam), dev(bet), y, mean, rstd, add=brd, x_sum=xs, add_dropout=ops.drop(p, seed, site))
    keep = torch.from_numpy(keep_mask(p, seed, site, rows * d)).view(rows, d)
    ref_sum = x.double() + brd.float().cpu().double() * keep / (1 - p)
    close(xs, ref_sum, 1e-6, 1e-6)
    close(y, O.layer_norm(ref_sum, gam.double(), bet.double()), 2e-3, 2e-3)
    # backward with the dropped copy of dx in fp16
    dy = (rnd(rows, d, seed=5) * 0.1).half()
    dx = torch.empty(rows, d, device=DEV); dxd = torch.empty(rows, d, dtype=H16, device=DEV)
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    ws = torch.empty(ops.layernorm_bwd_ws(rows, d), device=DEV)
    ops.layernorm_bwd(dev(dy), xs, dev(gam), mean, rstd, dx, dg, db, ws, dx_drop=dxd, dropout=ops.drop(p, seed + 1, site))
    xr = ref_sum.clone().requires_grad_(True); gr = gam.double().requires_grad_(True); brr = bet.double().requires_grad_(True)
    O.layer_norm(xr, gr, brr).backward(dy.double())
    close(dx, xr.grad, 1e-4, 1e-5)
    close(dg, gr.grad, 1e-4, 1e-4); close(db, brr.grad, 1e-4, 1e-4)
    keep2 = torch.from_numpy(keep_mask(p, seed + 1, site, rows * d)).view(rows, d)
    close(dxd, xr.grad * keep2 / (1 - p), 2e-3, 1e-5)


def test_convert_cast_and_ce_scale(ops):
    x = rnd(70, 96, seed=1)
    h = ops.convert(dev(x), torch.empty(70, 96, dtype=H16, device=DEV))
    assert torch.equal(h.cpu(), x.half())                                   # round to nearest even, as torch
    back = ops.convert(h, torch.empty(70, 96, device=DEV))
    assert torch.equal(back.cpu(), x.half().float())
    w = rnd(64, 96, seed=2)
    wd, wt = torch.empty(64, 96, dtype=H16, device=DEV), torch.empty(96, 64, dtype=H16, device=DEV)
    ops.cast_weights(dev(w), wd, wt)
    assert torch.equal(wd.cpu(), w.half()) and torch.equal(wt.cpu(), w.half().T)
    # CE backward multiplies by the device-resident loss scale
    rows, V = 50, 26
    logits, labels = rnd(rows, V, seed=3), torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(4))
    labels[::7] = -100
    lse = torch.empty(rows, device=DEV); am = torch.empty(rows, dtype=torch.int64, device=DEV); stats = torch.zeros(2, device=DEV)
    ops.ce_fwd(dev(logits), dev(labels), lse, am, stats)
    scale = torch.tensor([1024.0, 0, 0, 0], device=DEV)
    d1, d2 = torch.empty(rows, V, dtype=H16, device=DEV), torch.empty(rows, V, device=DEV)
    ops.ce_bwd(dev(logits), dev(labels), lse, stats, 0.25, d1, scale_dev=scale)
    ops.ce_bwd(dev(logits), dev(labels), lse, stats, 0.25, d2)
    close(d1, d2.cpu().double() * 1024.0, 2e-3, 1e-6)


@pytest.mark.parametrize("form", [107, 108])
@pytest.mark.parametrize("dt", [H16, torch.bfloat16])
def test_wgrad_leaves_out_padded_token_blocks(ops, dt, form):
    """afm_gemm_desc.k_live: 64-token blocks whose dy rows are exact zeros (padded positions of a training step) are left out of
    the token axis -- same weight and bias gradients as the full sweep, one by one and grouped; in the eight-wave unit (107) and in the
    four-wave unit of round 5 (108, csrc/afm_gemm_tnw4_impl.h)."""
    R = 8192
    live = torch.ones(R // 64, dtype=torch.uint8)
    live[5:40] = 0; live[77] = 0; live[100:] = 0                 # long dead runs, a single dead block, a dead tail
    rows_live = live.repeat_interleave(64).bool()
    shapes = [(512, 512), (1536, 512), (512, 2048)]
    ten, descs_a, descs_b, outs_a, outs_b = [], [], [], [], []
    for i, (M, N) in enumerate(shapes):
        dy = (rnd(R, M, seed=3 + i) * 0.5); dy[~rows_live] = 0.0
        a, b = dev(dy, dt), dev(rnd(R, N, seed=13 + i), dt)
        ten += [a, b]
        for descs, outs, kl in ((descs_a, outs_a, None), (descs_b, outs_b, dev(live))):
            g, gb = torch.zeros(M, N, device=DEV), torch.zeros(M, device=DEV)
            descs.append(ops.gemm_desc(a, b, g, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb, k_live=kl, variant=form)); outs.append((g, gb))
            ten.append(kl)
        g1, gb1 = torch.zeros(M, N, device=DEV), torch.zeros(M, device=DEV)
        ops.gemm(a, b, g1, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb1, k_live=dev(live), variant=form)   # alone, 256 x 256 tiles
        assert ("w4" in ops.last_algo()) == (form == 108)
        ref = a.double().T @ b.double()
        close(g1, ref.cpu(), 1e-4, 2e-4 * math.sqrt(R) / 4)
        close(gb1, a.double().sum(0).cpu(), 1e-4, 2e-4 * math.sqrt(R) / 4)
    ops.gemm_group(descs_a); ops.gemm_group(descs_b)
    assert ops.last_algo() == ("mfma_tn_groupw4" if form == 108 else "mfma_tn_group256")
    for (g0, b0), (g1, b1) in zip(outs_a, outs_b):
        torch.testing.assert_close(g0, g1, rtol=1e-5, atol=2e-4)
        torch.testing.assert_close(b0, b1, rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("N,K,act", [(512, 512, 0), (512, 2048, 0), (2048, 512, 5), (1536, 512, 0)])
def test_nt_gemm_writes_padded_row_tiles_as_zeros(ops, N, K, act):
    """afm_gemm_desc.k_live on the NT form (dgrad): row tiles of A that are nothing but padded positions (zero rows) come out as
    zero rows without being computed -- the same C as the full product, through the loader-wave and the 256 x 256 kernels."""
    M = 4096
    live = torch.ones(M // 64, dtype=torch.uint8)
    live[4:12] = 0; live[13] = 0; live[20:44] = 0; live[60:] = 0       # whole dead 256-row tiles, a tile with one dead block, a dead tail
    rl = live.repeat_interleave(64).bool()
    a = rnd(M, K, seed=1) * 0.5; a[~rl] = 0.0
    ad, wd = dev(a, H16), dev(rnd(N, K, seed=2) * 0.1, H16)
    pre = dev(rnd(M, N, seed=3), H16) if act == 5 else None
    outs = []
    for hint in (None, dev(live)):
        c = torch.full((M, N), 3.0, dtype=H16, device=DEV)
        ops.gemm(ad, wd, c, act=act, pre_act=pre, k_live=hint)
        assert ops.last_algo() in ("mfma_nt", "mfma_nt_256", "mfma_nt_pp", "mfma_nt_w4")
        outs.append(c)
    assert torch.equal(outs[0], outs[1])
    assert float(outs[1][~rl.to(DEV)].abs().max()) == 0.0


def test_gated_dgrad_writes_padded_row_tiles_as_zeros(ops):
    """The same hint on the gated FFN's dgrad form (act GLU_BWD: C is M x 2f, [dg * saved_a | dg * saved_b] interleaved)."""
    M, f, d = 2048, 256, 512
    live = torch.ones(M // 64, dtype=torch.uint8); live[4:8] = 0; live[9] = 0; live[16:] = 0
    rl = live.repeat_interleave(64).bool()
    dy = rnd(M, d, seed=1) * 0.5; dy[~rl] = 0.0
    dyd, w2t = dev(dy, H16), dev(rnd(f, d, seed=2) * 0.1, H16)          # (f x d): the NT operand of dg = dy W2
    saved = dev(rnd(M, 2 * f, seed=3), H16)
    outs = []
    for hint in (None, dev(live)):
        c = torch.full((M, 2 * f), 3.0, dtype=H16, device=DEV)
        ops.gemm(dyd, w2t, c, act=8, pre_act=saved, glu_rows=f, algo=2, k_live=hint)
        assert "glu" in ops.last_algo()
        outs.append(c)
    assert torch.equal(outs[0], outs[1])
    assert float(outs[1][~rl.to(DEV)].abs().max()) == 0.0


@pytest.mark.parametrize("dt", [H16, torch.float32])
def test_layernorm_backward_skips_padded_row_blocks(ops, dt):
    """afm_ln_shape.row_live: blocks of 64 rows with zero dy / dres are written as zeros without being read: same dx, dropped copy,
    dgamma and dbeta as the full pass."""
    rows, d = 1024, 512
    live = torch.ones(rows // 64, dtype=torch.uint8); live[3:9] = 0; live[12:] = 0
    rl = live.repeat_interleave(64).bool()
    x = rnd(rows, d, seed=1) * 2 + 0.3
    gam = 1 + 0.1 * rnd(d, seed=2)
    dy = rnd(rows, d, seed=3); dy[~rl] = 0.0
    dres = rnd(rows, d, seed=4); dres[~rl] = 0.0
    mean = dev(x.mean(1)); rstd = dev(1.0 / (x.var(1, unbiased=False) + 1e-5).sqrt())
    res = []
    for hint in (None, dev(live)):
        dx = torch.full((rows, d), 9.0, device=DEV); dxd = torch.full((rows, d), 9.0, dtype=dt, device=DEV)
        dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
        ws = torch.empty(ops.layernorm_bwd_ws(rows, d), device=DEV)
        ops.layernorm_bwd(dev(dy, dt), dev(x), dev(gam), mean, rstd, dx, dg, db, ws, dres=dev(dres), dx_drop=dxd,
                          dropout=ops.drop(0.1, 5, 6), row_live=hint)
        res.append((dx, dxd, dg, db))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    torch.testing.assert_close(res[0][2], res[1][2], rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(res[0][3], res[1][3], rtol=1e-5, atol=1e-4)
    assert float(res[1][0][~rl.to(DEV)].abs().max()) == 0.0


# ------------------------------------------------------------------ loss scaler
def test_adam_with_loss_scaler_skips_and_rescales(ops):
    """GradScaler semantics on the device: gradients arrive S times too large; a non-finite norm skips the step (parameters and
    moments untouched, gradients zeroed) and halves S; `interval` good steps in a row double it; bias corrections count the
    steps TAKEN."""
    n = 4096 + 3
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2) * 0.01
    S = 1024.0
    pd = dev(p0).clone(); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    shadow = torch.empty(n, dtype=H16, device=DEV)
    st = torch.tensor([S, 0.0, 0.0, 0.0], device=DEV)
    pr, mr, vr = p0.clone().double(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    ss = torch.zeros(1, device=DEV)
    taken = 0
    for it in range(6):
        lr, b1 = O.onecycle(it, 10, 1e-2)
        scale_now = float(st[0])
        gd = dev(g0) * scale_now * (it + 1)
        if it in (1, 4):
            gd[17] = float("inf")                                           # fp16 overflow somewhere in the backward pass
        hyper = torch.tensor([lr, b1, 0.999, 1e-8, 0.01, -1.0, -1.0, 1.0, 1.0, 1.0], device=DEV)   # bc fields unused with a scaler
        ss.zero_(); ops.sumsq(gd, ss)
        before = pd.clone()
        ops.adam_step(pd, gd, m, v, hyper, ss, shadow, zero_grad=True, scaler=st)
        ops.scaler_update(st, ss, growth=2.0, backoff=0.5, interval=2)
        assert float(gd.abs().max()) == 0.0
        if it in (1, 4):
            assert torch.equal(pd, before)                                  # skipped
        else:
            taken += 1
            gt = g0.double() * (it + 1)
            coef = min(1.0, 1.0 / (float(gt.norm()) + 1e-6))
            O.adam_step(pr, gt * coef, mr, vr, taken, lr, b1, 0.999, 1e-8, 0.01, True)
            close(pd, pr, 2e-5, 2e-6, f"step {it}")
            close(shadow, pd.float().cpu().half(), 0, 0)
    # scale history: 1024 -(good)-> tracker 1 -(inf)-> 512 -(good, good)-> 1024 -(inf)-> 512 -(good)-> 512
    assert st.cpu().tolist() == [512.0, 1.0, 4.0, 2.0]


# ------------------------------------------------------------------ training loop vs the reference goldens
@pytest.mark.parametrize("name", ["model_plain", "model_gated_learned", "model_postln_relu", "model_postln_gated"])
def test_f16_two_optimizer_steps_vs_reference_golden(name):
    """fp16 forward / backward with the dynamic loss scale against the reference's fp32 run: logits inside the north star's 1e-3,
    parameters after two optimiser steps (accumulate 4, clip, AdamW + OneCycle) at fp16-gradient tolerance."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodalanalytical_amd.engine import Seq2SeqEngine
    from multimodalanalytical_amd.optim import FusedAdamOneCycle
    t = G.load(name); cfg = G.model_cfg(t["meta"]); m = t["meta"]
    eng = Seq2SeqEngine(dict(cfg), m["data_config"], "Smiles", m["data_config"]["Smiles"]["vocab_size"], device=DEV, compute_dtype=H16)
    eng.load_state_dict(t["sd"])
    opt = FusedAdamOneCycle(eng, m["optimiser"], lr=m["lr"], weight_decay=m["weight_decay"], num_steps=m["total_steps"], clip_grad=m["clip"])

    def inputs(i):
        enc, am, dec, dm, labels = O.batch_to_model_inputs(G.batch_of(t, i), "Smiles")
        to = lambda x: {k: to(v) for k, v in x.items()} if isinstance(x, dict) else x.to(DEV)
        return to(enc), am.to(DEV), dec.to(DEV), dm.to(DEV), labels.to(DEV)
    for step in (1, 2):
        for i in range(4):
            out = eng.forward(*inputs(i), backward=True, loss_scale=1.0 / m["acc_batches"])
            if step == 1:
                ref = t[f"b{i}"]
                err = float((out["logits"].cpu().double() - ref["logits"].double()).abs().max() / ref["logits"].double().abs().max())
                assert err < 1e-3, (i, err)
        opt.step()
        torch.testing.assert_close(opt.grad_norm().cpu(), t[f"step{step}"]["grad_norm"], rtol=5e-3, atol=1e-6)
        # Adam normalises the step (after step 1 every element has moved by ~lr whatever its gradient's size), so the measure
        # is the error of the MOVEMENT over all parameters; elements whose gradient is at the fp16 noise floor may go the other way
        num = den = 0.0
        for k, ref in t[f"step{step}"].items():
            if k == "grad_norm" or k.endswith("in_proj_bias"):
                continue
            got = eng.ps.p(k).cpu().double()
            num += float((got - ref.double()).norm()) ** 2
            den += float((ref.double() - t["sd"][k].double()).norm()) ** 2
        assert (num / den) ** 0.5 < 0.15, (step, (num / den) ** 0.5)
    assert eng.scaler.cpu().tolist() == [65536.0, 2.0, 2.0, 0.0]
