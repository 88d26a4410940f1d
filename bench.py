#!/usr/bin/env python3
"""Headline benchmark: train samples/sec of the spectra->SMILES path on N MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 (or --force-ddp) without a launcher environment: bench.py starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
itself as a CHILD process before anything touches the GPU, relays the job's one JSON line and exits with its status; under
torch.distributed.run (RANK set) it is one rank of the job.  WORLD_SIZE != --gpus is an error in every case.

A "step" is one optimiser step of the reference's training configuration: acc_batches (4) micro-batches of
`batch` (128) samples each through HFWrapper.training_step (forward + backward, dropout 0.1 active), then
gradient clip 1.0 + AdamW + OneCycleLR; with N > 1 the flat gradient buffer is all-reduced over RCCL once per
step, overlapped with the last backward.  Inputs are synthetic (multimodalanalytical_amd/synth.py, seeded): the 177 000-sample
set of SURVEY 8(d) is resident in HBM before the timed region as RAW pre-tokenised data, and every micro-batch is drawn, standardised,
patchified and collated on the device inside it (--input-path fixed: the pre-collated batches of rounds 1-3).  Prints ONE JSON line
(rank 0).

Precision modes, all timed in the same run and listed under `modes`:
  fp16          (`value`) fp16 operands, ONE MFMA pass per product forward and backward, fp32 accumulation / residual stream /
                statistics / master weights, dynamic loss scaling on the device: the reference's own GPU arithmetic
                (trainer/trainer.py:69, Lightning "16-mixed").  Logits 4e-4 .. 7e-4 of the CPU reference at c1..c5 and at the
                timed size (bar 1e-3; tests/test_gpu_shapes.py, tests/test_gpu_model.py), gradients 6e-4 .. 9e-4
  bf16x3-mixed  the forward on split bf16 operand pairs (three bf16 MFMA passes per product: logits within 1e-5), the backward
                on the single-pass bf16 kernels reading the hi planes of the saved pair tensors (gradients 4e-3)
  bf16x3        split pairs in both directions (gradients 1e-5 against the CPU reference)
  bf16          single bf16 pass everywhere (3e-3..6e-3 on the logits: outside the parity bar, for comparison only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16 peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
METRIC = "train samples/sec (IR+NMR→SMILES, enc1024/dec128) at 1/2/4/8 MI355X"    # BASELINE.json `metric`


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", help="c1..c5 (multimodalanalytical_amd/synth.py)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU micro-batch (default: the workload's, 128)")
    ap.add_argument("--acc", type=int, default=4)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16x3", "bf16x3-mixed", "bf16", "fp32"], help="mode of `value`")
    ap.add_argument("--other-modes", default="bf16x3-mixed,bf16x3,bf16", help="comma list of further modes timed in the same run ('' = none)")
    ap.add_argument("--other-steps", type=int, default=5)
    ap.add_argument("--extra-steps", type=int, default=10, help="timed steps of each --extra-workloads entry (5 for c4 / c5)")
    ap.add_argument("--extra-workloads", default="c3,c4,c5", help="comma list: further workloads timed in the primary mode (N = 1 only)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run logits / ids check against the CPU oracle")
    ap.add_argument("--input-path", default="loader", choices=["loader", "fixed"],
                    help="loader: every micro-batch is drawn inside the timed region from the resident 177 000-sample synthetic shard "
                         "(ShardLoader -> DeviceCollator -> afm_patch_preprocess; mixture workloads: MixtureLoader -> afm_mix_spectra); "
                         "fixed: 4 pre-collated batches per rank, cycled (the loop of rounds 1-3)")
    ap.add_argument("--no-input-compare", action="store_true", help="skip the fixed-batch comparison run (profiling: keeps the trace to the timed loop)")
    ap.add_argument("--set-size", type=int, default=177000, help="samples of the resident synthetic set (SURVEY 8d: synth-177K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-eval", action="store_true", help="skip the eval / decode section (greedy at the main workload, beam 5 at c5)")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=32, help="torch CPU threads of the baseline (more than 32 ran SLOWER on the GPU box; the host's core count is printed beside it)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="run the RCCL gradient exchange even with one rank (exercises the N>1 code path on a 1-GPU box)")
    return ap.parse_args()


def compute_dtype(name):
    from multimodalanalytical_amd.x2 import X2
    return {"fp16": torch.float16, "bf16": torch.bfloat16, "bf16x3": X2.dtype, "bf16x3-mixed": X2.dtype, "fp32": torch.float32}[name]


def backward_dtype(name):
    import torch
    return torch.bfloat16 if name == "bf16x3-mixed" else None


def time_kernel(fn, iters=20, warm=30):
    """Average device time of fn() in ms: HIP events on the stream the kernel is launched on (torch's current
    stream, which is the stream ops.py hands to the C ABI).  30 warm-up launches: a kernel timed right after an idle
    period reads ~15 % slow (clocks), and the committed rocprofv3 averages (profiles/) are taken warm."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


# ------------------------------------------------------------------------------------------------ roofline
def _rand(rows, cols, cd, dev, scale=1.0):
    from multimodalanalytical_amd import ops
    x = torch.randn(rows, cols, device=dev) * scale
    if cd == torch.float32:
        return x
    return ops.convert(x, ops.empty(rows, cols, cd, dev))


def kernel_rooflines(model, wl, B, mode, wl_name="c2"):
    """Live timings of the kernels that make up the step, at the workload's encoder shapes (the encoder is ~80 % of
    the FLOPs): each entry = one launch of one kernel.  `achieved` = ALGORITHMIC FLOPs of that launch / its average
    duration; algorithmic FLOPs count every matrix product the kernel has to evaluate ONCE at 2 FLOPs per
    multiply-add, whatever the number of MFMA passes the precision mode spends on it (bf16x3: three), see DESIGN.md
    section 4.  `mfma_frac_executed` prices the executed MFMA passes instead (how busy the matrix cores are).
    The entry with the largest share of the step becomes `roofline`; the others are listed in `roofline_kernels`."""
    from multimodalanalytical_amd import ops
    from multimodalanalytical_amd.lib import ACT_GELU_SAVE_GRAD, ACT_MUL_SAVED
    from multimodalanalytical_amd.x2 import X2
    cfg = wl["cfg"]
    eng = model.hf_model.engine
    dev, cd = eng.dev, eng.cd
    mixed = getattr(eng, "mixed", False)     # bf16x3 forward kernels, single-pass bf16 backward kernels on the hi planes
    S = sum(v[0] if isinstance(v, tuple) else v for v in wl["lens"].values())
    d, H = cfg["d_model"], cfg["encoder_attention_heads"]
    f = cfg["encoder_ffn_dim"] * (2 if cfg["gated_linear"] else 1)
    Le = cfg["encoder_layers"]
    M, dh = B * S, d // H
    fmode = "bf16x3" if mixed else mode                       # arithmetic of the forward kernels
    bmode = "bf16" if mixed else mode                         # ... of the backward kernels
    PASS = {"bf16x3": 3, "bf16": 1, "fp16": 1, "fp32": 1}
    ESZ = {"bf16": 2, "fp16": 2, "bf16x3": 4, "fp32": 4}
    peak = PEAK_BF16_TFLOPS if mode != "fp32" else 157.3
    out = []

    def hb(t):                                                # what a backward kernel reads of a saved forward tensor
        return t.hi if (mixed and isinstance(t, X2)) else t

    def empty_b(rows, cols, like=None):                       # backward activation gradient (row stride of `like` in mixed mode)
        if mixed and like is not None:
            return torch.empty(rows, like.ld, dtype=torch.bfloat16, device=dev)[:, :cols]
        return ops.empty(rows, cols, torch.bfloat16 if mixed else cd, dev)

    def add(name, kern, ms, flops, alg_bytes, calls, note, kmode):
        ach = flops / (ms * 1e-3) / 1e12
        out.append({"bound": "mfma", "kernel": kern, "what": name, "arithmetic": kmode, "achieved": round(ach, 1), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "mfma_frac_executed": round(min(1.0, PASS[kmode] * ach / peak), 4) if mode != "fp32" else None,
                    "avg_launch_ms": round(ms, 4), "algorithmic_flops": flops, "algorithmic_bytes": alg_bytes,
                    "hbm_time_at_peak_ms": round(alg_bytes / (PEAK_HBM_GBS * 1e9) * 1e3, 4),
                    "launches_per_micro_batch": calls, "ms_per_micro_batch": round(ms * calls, 3), "counts": note,
                    "traffic": None})

    # --- attention, encoder self-attention shape (packed QKV addressed in place, real dropout stream)
    qkv = _rand(M, 3 * d, cd, dev)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    o = ops.empty(M, d, cd, dev)
    do = empty_b(M, d, like=o if mixed else None)
    ops.convert(torch.randn(M, d, device=dev) * 0.01, do)
    dqkv = empty_b(M, 3 * d)
    lse = torch.empty(B * H * S, device=dev)
    delta = torch.empty_like(lse)
    dr = ops.drop(cfg["dropout"], 1, 3)
    prod = 2.0 * B * H * S * S * dh                      # one S x S x dh product over the batch
    lq, lo, lg = ops._ld(qkv), ops._ld(o), ops._ld(dqkv)
    # the keep-bit tensor of the attention-probability dropout, as the engine attaches it in a training step
    bits = None
    if eng.keep_bits and mode != "fp32" and cfg["dropout"] > 0:
        bits = torch.zeros(ops.attn_drop_bits_words(B, H, S, S), dtype=torch.int64, device=dev)
    bits_bytes = 0 if bits is None else bits.numel() * 8

    def shape(res, dt):
        s = ops.attn_shape(B, H, S, S, dh, dt, lq, lq, lq, lo, None, False, dr)
        s.reserved = res
        return ops.attn_set_drop_bits(s, bits)
    bdt = torch.bfloat16 if mixed else cd
    s0, s1, s2 = shape(0, cd), shape(1, bdt), shape(2, bdt)
    ef, eb = ESZ[fmode], ESZ[bmode]
    ms = time_kernel(lambda: ops.attn_fwd(s0, q, k, v, o, lse))
    algo = ops.last_algo()
    add("attention forward (encoder self-attention)", f"afm_attn_fwd[{algo}]", ms, 2 * prod,
        ef * 4 * M * d + bits_bytes, Le, "2 products: Q K^T, P V" + ("; writes the dropout keep bits" if bits is not None else ""), fmode)
    qb, kb, vb, ob = hb(q), hb(k), hb(v), hb(o)
    ms = time_kernel(lambda: ops.attn_bwd(s1, qb, kb, vb, ob, do, lse, delta, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], lg, lg, lg))
    algo = ops.last_algo()
    add("attention backward, dQ kernel", f"afm_attn_bwd[{algo}] dQ", ms, 3 * prod, eb * 6 * M * d + bits_bytes, Le,
        "3 products: Q K^T and dO V^T recomputed, dS K" + ("; reads the dropout keep bits" if bits is not None else ""), bmode)
    ms = time_kernel(lambda: ops.attn_bwd(s2, qb, kb, vb, ob, do, lse, delta, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], lg, lg, lg))
    add("attention backward, dK/dV kernel", f"afm_attn_bwd[{algo}] dK/dV", ms, 4 * prod, eb * 6 * M * d + bits_bytes, Le,
        "4 products: Q K^T and dO V^T recomputed, P^T dO, dS^T Q" + ("; reads the dropout keep bits" if bits is not None else ""), bmode)

    # --- the decoder's cross-attention backward (T queries against the S memory positions): one fused kernel where the library has one
    # (round 6, csrc/afm_attn_fsq_impl.h: T <= 128, keep-bit dropout or none), the dQ / dK-dV pair elsewhere
    T = int(wl["T"])
    if not mixed and mode in ("fp16", "bf16"):
        Ld = cfg["decoder_layers"]
        qx = _rand(B * T, d, cd, dev)
        kvx = _rand(M, 2 * d, cd, dev)
        ox = ops.empty(B * T, d, cd, dev)
        dox = ops.empty(B * T, d, cd, dev)
        ops.convert(torch.randn(B * T, d, device=dev) * 0.01, dox)
        dqx, dkvx = ops.empty(B * T, d, cd, dev), ops.empty(M, 2 * d, cd, dev)
        lsex = torch.empty(B * H * T, device=dev)
        deltax = torch.empty_like(lsex)
        sx = ops.attn_shape(B, H, T, S, dh, cd, d, 2 * d, 2 * d, d, torch.zeros(B, S, dtype=torch.uint8, device=dev), False, dr)
        bitsx = None
        if eng.keep_bits and cfg["dropout"] > 0:
            bitsx = torch.zeros(ops.attn_drop_bits_words(B, H, T, S), dtype=torch.int64, device=dev)
            ops.attn_set_drop_bits(sx, bitsx)
        ops.attn_fwd(sx, qx, kvx[:, :d], kvx[:, d:], ox, lsex)
        sx.reserved |= 262144 if getattr(eng, "xattn_fused", False) else 0
        ms = time_kernel(lambda: ops.attn_bwd(sx, qx, kvx[:, :d], kvx[:, d:], ox, dox, lsex, deltax, dqx, dkvx[:, :d], dkvx[:, d:], d, 2 * d, 2 * d))
        algo = ops.last_algo()
        prodx = 2.0 * B * H * T * S * dh
        add("cross-attention backward (decoder queries x encoder memory), dQ + dK + dV", f"afm_attn_bwd[{algo}] cross", ms, 5 * prodx,
            ESZ[mode] * (4 * M * d + 4 * B * T * d) + (0 if bitsx is None else bitsx.numel() * 8), Ld,
            ("5 products in one kernel: K Q^T, V dO^T, K^T dS^T, dO^T P, Q^T dS" if algo == "attn_fsq" else "7 products in two kernels (S and dP twice)")
            + ("; reads the dropout keep bits" if bitsx is not None else ""), mode)

    # --- GEMMs of one encoder layer at their training epilogues
    x = _rand(M, d, cd, dev)
    pre = ops.empty(M, f, cd, dev)                       # stored keep*scale*GELU' (forward output, backward input)
    du = empty_b(M, f, like=pre if mixed else None)
    ops.convert(torch.randn(M, f, device=dev) * 0.01, du)
    gw = torch.zeros(f, d, device=dev)
    gb = torch.zeros(f, device=dev)
    xb = hb(x)
    ms = time_kernel(lambda: ops.gemm(du, xb, gw, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb))
    add("FFN up-projection weight gradient (bias gradient fused)", f"afm_gemm[{ops.last_algo()}] dW1 {f}x{d} over {M} tokens",
        ms, 2.0 * M * d * f, eb * M * (f + d) + 4 * f * d, 1 * Le, "1 product", bmode)
    wq = eng.W("encoder.layers.0.self_attn.in_proj_weight", 3 * d, d)
    bq = eng.ps.p("encoder.layers.0.self_attn.in_proj_bias")
    oq = ops.empty(M, 3 * d, cd, dev)
    ms = time_kernel(lambda: ops.gemm(x, wq, oq, trans_b=True, bias=bq))
    add("QKV projection forward", f"afm_gemm[{ops.last_algo()}] {M}x{3 * d}x{d}", ms, 2.0 * M * 3 * d * d,
        ef * (M * d + 3 * d * d + M * 3 * d), Le, "1 product", fmode)
    if not cfg["gated_linear"]:
        w1 = eng.W("encoder.layers.0.linear1.weight", f, d)
        b1 = eng.ps.p("encoder.layers.0.linear1.bias")
        g = ops.empty(M, f, cd, dev)
        ms = time_kernel(lambda: ops.gemm(x, w1, g, trans_b=True, bias=b1, act=ACT_GELU_SAVE_GRAD, pre_act=pre, dropout=dr))
        by = ef * (M * d + f * d) + 2 * ef * M * f
        add("FFN up-projection forward (bias + GELU + dropout fused, keep*scale*GELU' stored)",
            f"afm_gemm[{ops.last_algo()}] {M}x{f}x{d}", ms, 2.0 * M * d * f, by, Le, "1 product; HBM-heavy: two M x f outputs", fmode)
        if by / (PEAK_HBM_GBS * 1e9) > 2.0 * M * d * f * PASS[fmode] / (peak * 1e12):
            out[-1].update(bound="hbm", achieved=round(by / (ms * 1e-3) / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                           frac=round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4))
        if mode != "fp32":
            w2t = hb(eng.wt["encoder.layers.0.linear2.weight"])      # (f x d): dgrad of linear2 as an NT GEMM
            dy = ops.empty(M, d, torch.bfloat16 if mixed else cd, dev)
            ops.convert(torch.randn(M, d, device=dev) * 0.01, dy)
            preb = hb(pre)
            ms = time_kernel(lambda: ops.gemm(dy, w2t, du, trans_b=True, act=ACT_MUL_SAVED, pre_act=preb))
            add("FFN down-projection data gradient (x stored keep*scale*GELU')", f"afm_gemm[{ops.last_algo()}] {M}x{f}x{d}",
                ms, 2.0 * M * d * f, eb * (M * d + f * d + 2 * M * f), Le, "1 product", bmode)
    if mode != "fp32" and not mixed:
        from multimodalanalytical_amd.lib import ACT_GLU_BWD, ACT_GLU_SAVE
        fh = cfg["encoder_ffn_dim"]                          # hidden width (the gated forms carry 2 fh columns through the up-projection)
        # --- the long-K plain products (bias only): FFN down-projection forward, QKV data gradient
        gact = _rand(M, fh, cd, dev)
        w2 = eng.W("encoder.layers.0.linear2.weight", d, fh)
        b2 = eng.ps.p("encoder.layers.0.linear2.bias")
        br = ops.empty(M, d, cd, dev)
        ms = time_kernel(lambda: ops.gemm(gact, w2, br, trans_b=True, bias=b2))
        add("FFN down-projection forward", f"afm_gemm[{ops.last_algo()}] {M}x{d}x{fh}", ms, 2.0 * M * d * fh,
            ef * (M * fh + d * fh + M * d), Le, "1 product", fmode)
        wqt = eng.wt["encoder.layers.0.self_attn.in_proj_weight"]          # (d x 3d): dgrad of the packed projection as an NT GEMM
        dx = ops.empty(M, d, cd, dev)
        ms = time_kernel(lambda: ops.gemm(dqkv, wqt, dx, trans_b=True))
        add("QKV projection data gradient", f"afm_gemm[{ops.last_algo()}] {M}x{d}x{3 * d}", ms, 2.0 * M * d * 3 * d,
            eb * (M * 3 * d + 3 * d * d + M * d), Le, "1 product", bmode)
        # --- an encoder layer's weight gradients as the engine launches them: ONE grouped launch (afm_gemm_group)
        dy = ops.empty(M, d, cd, dev)
        ops.convert(torch.randn(M, d, device=dev) * 0.01, dy)
        k2 = 2 if cfg["gated_linear"] else 1
        duv = ops.empty(M, k2 * fh, cd, dev)
        ops.convert(torch.randn(M, k2 * fh, device=dev) * 0.01, duv)
        p0 = "encoder.layers.0."
        G, gv = eng.G, lambda n, a0, a1: eng.ps.vec_span(eng.ps.grad, n, a0, a1)
        kwg = dict(trans_a=True, trans_b=False, accumulate=True)
        descs = [ops.gemm_desc(dy, gact, G(p0 + "linear2.weight", d, fh), a_colsum=gv(p0 + "linear2.bias", 0, d), **kwg),
                 ops.gemm_desc(duv, x, G(p0 + "linear1.weight", k2 * fh, d), a_colsum=gv(p0 + "linear1.bias", 0, k2 * fh),
                               glu_rows=fh if cfg["gated_linear"] else 0, **kwg),
                 ops.gemm_desc(dy, o, G(p0 + "self_attn.out_proj.weight", d, d), a_colsum=gv(p0 + "self_attn.out_proj.bias", 0, d), **kwg),
                 ops.gemm_desc(dqkv, x, G(p0 + "self_attn.in_proj_weight", 3 * d, d), a_colsum=gv(p0 + "self_attn.in_proj_bias", 0, 3 * d), **kwg)]
        ms = time_kernel(lambda: ops.gemm_group(descs))
        wflops = 2.0 * M * (d * fh + k2 * fh * d + d * d + 3 * d * d)
        add("weight gradients of one encoder layer, grouped launch (bias gradients fused)", f"afm_gemm_group[{ops.last_algo()}] 4 problems over {M} tokens",
            ms, wflops, eb * M * (2 * d + fh + k2 * fh + d + 3 * d + d) + 4 * (d * fh + k2 * fh * d + 4 * d * d), Le, "4 products", bmode)
        eng.ps.grad.zero_()
        if cfg["gated_linear"]:
            # --- gated FFN: gelu(u) * v, dropout and the stored factors in the up-projection's epilogue; [du | dv] in the dgrad epilogue
            wg = eng.w_glu[p0 + "linear1.weight"]
            bg = eng.ps.vec_span(eng.ps.flat, p0 + "linear1.bias", 0, 2 * fh)
            uv = ops.empty(M, 2 * fh, cd, dev)
            gg = ops.empty(M, fh, cd, dev)
            ms = time_kernel(lambda: ops.gemm(x, wg, gg, trans_b=True, bias=bg, act=ACT_GLU_SAVE, pre_act=uv, dropout=dr, glu_rows=fh))
            by = ef * (M * d + 2 * fh * d + M * fh + 2 * M * fh)
            add("gated FFN up-projection forward (bias + gelu(u) v + dropout fused, keep*scale*[gelu'(u) v | gelu(u)] stored)",
                f"afm_gemm[{ops.last_algo()}] {M}x{2 * fh}x{d}", ms, 2.0 * M * d * 2 * fh, by, Le, "1 product; three M x f outputs", fmode)
            w2t = eng.wt[p0 + "linear2.weight"]                # (fh x d)
            ms = time_kernel(lambda: ops.gemm(dy, w2t, duv, trans_b=True, act=ACT_GLU_BWD, pre_act=uv, glu_rows=fh))
            add("gated FFN down-projection data gradient (x stored factors -> [du | dv])", f"afm_gemm[{ops.last_algo()}] {M}x{fh}x{d}",
                ms, 2.0 * M * d * fh, eb * (M * d + fh * d + 4 * M * fh), Le, "1 product", bmode)
            wgt = eng.wt_glu[p0 + "linear1.weight"]            # (d x 2 fh)
            dh = ops.empty(M, d, cd, dev)
            ms = time_kernel(lambda: ops.gemm(duv, wgt, dh, trans_b=True))
            add("gated FFN up-projection data gradient", f"afm_gemm[{ops.last_algo()}] {M}x{d}x{2 * fh}", ms, 2.0 * M * d * 2 * fh,
                eb * (M * 2 * fh + 2 * fh * d + M * d), Le, "1 product", bmode)
    # --- HBM-bound kernels of the step (LayerNorm / AdamW on the HBM roofline): bytes every launch must move
    def add_hbm(name, kern, ms, alg_bytes, calls, note):
        gbs = alg_bytes / (ms * 1e-3) / 1e9
        out.append({"bound": "hbm", "kernel": kern, "what": name, "arithmetic": "fp32 statistics / " + mode, "achieved": round(gbs, 1),
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "mfma_frac_executed": None,
                    "avg_launch_ms": round(ms, 4), "algorithmic_flops": 0.0, "algorithmic_bytes": alg_bytes,
                    "hbm_time_at_peak_ms": round(alg_bytes / (PEAK_HBM_GBS * 1e9) * 1e3, 4),
                    "launches_per_micro_batch": calls, "ms_per_micro_batch": round(ms * calls, 3), "counts": note, "traffic": None})
    if mode != "fp32" and not mixed:
        xs = torch.randn(M, d, device=dev)
        branch = _rand(M, d, cd, dev)
        gam, bet = eng.ps.p("encoder.layers.0.norm1.weight"), eng.ps.p("encoder.layers.0.norm1.bias")
        yln, xsum = ops.empty(M, d, cd, dev), torch.empty(M, d, device=dev)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        ms = time_kernel(lambda: ops.layernorm_fwd(xs, gam, bet, yln, mean, rstd, add=branch, x_sum=xsum, add_dropout=dr))
        add_hbm("LayerNorm forward with the residual add and the branch dropout fused (encoder rows)", f"afm_layernorm_fwd {M}x{d}", ms,
                M * d * (4 + ef + 4 + ef) + 8 * M, 2 * Le + 1, "reads the fp32 stream and the 16-bit branch, writes the summed stream (fp32) and the operand (16-bit)")
        dyl = _rand(M, d, cd, dev, 0.01)
        dres, dxl, dxd = torch.randn(M, d, device=dev) * 0.01, torch.empty(M, d, device=dev), ops.empty(M, d, cd, dev)
        dgam, dbet = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        ws = torch.empty(max(1, ops.layernorm_bwd_ws(M, d)), device=dev)
        ms = time_kernel(lambda: ops.layernorm_bwd(dyl, xsum, gam, mean, rstd, dxl, dgam, dbet, ws, dres=dres, dx_drop=dxd, dropout=dr))
        add_hbm("LayerNorm backward with the residual-gradient add and the dropped copy for the preceding branch fused (encoder rows)",
                f"afm_layernorm_bwd {M}x{d}", ms, M * d * (eb + 4 + 4 + 4 + eb) + 8 * M, 2 * Le + 1,
                "reads dy (16-bit), the saved stream and the residual gradient (fp32), writes dx (fp32) and dropout'(dx) (16-bit)")
        del xs, branch, yln, xsum, dyl, dres, dxl, dxd
        # Adam(W) over the flat parameter buffer, on copies (the weights of the run stay as trained)
        ps = eng.ps
        pc, gc, mc, vc = ps.flat.clone(), torch.randn_like(ps.flat) * 1e-3, ps.exp_avg.clone(), ps.exp_avg_sq.clone()
        lowc = ps.bf16.clone() if ps.bf16 is not None else None
        hyper = torch.tensor([1e-4, 0.9, 0.999, 1e-8, 0.0, 0.1, 0.001, 1.0, 1.0, 1.0], device=dev)
        ssq = torch.ones(1, device=dev)
        ms = time_kernel(lambda: ops.adam_step(pc, gc, mc, vc, hyper, ssq, lowc, zero_grad=True, scaler=None), iters=10, warm=5)
        add_hbm("AdamW step over the flat parameter buffer (clip factor, 16-bit weight shadow and gradient zeroing fused)",
                f"afm_adam_step {pc.numel()} parameters", ms, pc.numel() * (16 + 12 + 4 + (2 if lowc is not None else 0)), 0,
                "per optimiser step, not per micro-batch: reads p, g, m, v; writes p, m, v, zeroes g, writes the 16-bit shadow")
        out[-1]["launches_per_step"] = 1
        del pc, gc, mc, vc, lowc
        # --- decoder-row products (M = B * T rows: 64 row tiles of 256 on 256 CUs)
        T = wl["T"]
        Mt = B * T
        Ld = cfg["decoder_layers"]
        fd = cfg["decoder_ffn_dim"]
        xt = _rand(Mt, d, cd, dev)
        for (nm, N_, K_, wname, calls) in (("decoder self-attention QKV projection forward", 3 * d, d, "decoder.layers.0.self_attn.in_proj_weight", Ld),
                                          ("decoder output projection forward (self- and cross-attention)", d, d, "decoder.layers.0.self_attn.out_proj.weight", 2 * Ld)):
            wdec = eng.W(wname, N_, K_)
            odec = ops.empty(Mt, N_, cd, dev)
            bdec = eng.ps.p(wname.replace("weight", "bias"))
            ms = time_kernel(lambda: ops.gemm(xt, wdec, odec, trans_b=True, bias=bdec))
            add(nm, f"afm_gemm[{ops.last_algo()}] {Mt}x{N_}x{K_}", ms, 2.0 * Mt * N_ * K_, ef * (Mt * K_ + N_ * K_ + Mt * N_), calls, "1 product", fmode)
        if not cfg["gated_linear"]:
            gdec = _rand(Mt, fd, cd, dev)
            w2d, b2d = eng.W("decoder.layers.0.linear2.weight", d, fd), eng.ps.p("decoder.layers.0.linear2.bias")
            odec = ops.empty(Mt, d, cd, dev)
            ms = time_kernel(lambda: ops.gemm(gdec, w2d, odec, trans_b=True, bias=b2d))
            add("decoder FFN down-projection forward", f"afm_gemm[{ops.last_algo()}] {Mt}x{d}x{fd}", ms, 2.0 * Mt * d * fd,
                ef * (Mt * fd + d * fd + Mt * d), Ld, "1 product", fmode)
    # traffic from committed PMC passes of the same launches (profiles/r02_*_pmc.json: {kernel what: bytes})
    if B == 128 and S == 1024:
        tables = {}
        for e in out:
            km = e["arithmetic"]
            if km not in tables:
                tables[km] = {}
                old = (f"r04_{km}_pmc.json", f"r03_{km}_pmc.json", f"r02_{km}_pmc.json") if wl_name == "c2" else ()
                for cand in (f"r05_{wl_name}_{km}_pmc.json",) + old:   # PMC passes of the same launches, newest round first
                    pmc = os.path.join(ROOT, "profiles", cand)
                    if os.path.exists(pmc):
                        tables[km] = json.load(open(pmc))
                        break
            e["traffic"] = tables[km].get(e["what"], {}).get("hbm_bytes_per_launch")
    out.sort(key=lambda e: -e["ms_per_micro_batch"])
    return out


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(model, wl, name, cpu_batch, steps, threads):
    """the reference path on this host's cores, fp32, same workload shapes: one warm-up step, then
    `steps` timed steps (forward + backward + clip + AdamW on a micro-batch of `cpu_batch` samples), for
      kind "port"      the op-by-op oracle (oracle/afm_oracle.py), pinned to the reference's golden vectors, and
      stock_torch      the same layer stack wired from torch.nn.TransformerEncoder/Decoder modules the way the
                       reference wires them (oracle/stock_torch.py): the reference's own arithmetic."""
    from multimodalanalytical_amd import synth
    from oracle import afm_oracle as O
    from oracle import stock_torch as ST
    host_cores = os.cpu_count() or 1
    cores = max(1, min(threads or host_cores, host_cores))
    torch.set_num_threads(cores)
    eng = model.hf_model.engine
    sd = {k: v.detach().float().cpu() for k, v in eng.state_dict().items() if not k.startswith("decoder.embedding.")}
    batch, _ = synth.make_batch(name, cpu_batch, seed=99)
    enc, am, dec, dm, labels = O.batch_to_model_inputs(batch, "Smiles")
    cfg = dict(wl["cfg"]); cfg["dropout"] = 0.0
    tr = O.OracleTrainer(sd, cfg, wl["data"], "Smiles", lr=1e-4, total_steps=steps + 2, acc_batches=1)
    tr.micro_batch(enc, am, dec, dm, labels)                       # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.micro_batch(enc, am, dec, dm, labels)
    dt = (time.perf_counter() - t0) / steps
    port = {"value": round(cpu_batch / dt, 4), "unit": "samples/s", "cores": cores, "host_cores": host_cores,
            "sample": f"op-by-op oracle (oracle/afm_oracle.py, the parity checker; it materialises the S x S scores), {steps} timed "
                      f"steps after 1 warm-up, {dt:.2f} s/step"}
    res = dict(port, kind="port", sample=f"{steps} timed optimiser steps (after 1 warm-up) on a micro-batch of {cpu_batch} samples of "
                                         f"workload {name}: fwd+bwd+clip+AdamW, fp32, {dt:.2f} s/step; " + port["sample"])
    if not cfg["gated_linear"]:
        m = ST.StockSeq2Seq(cfg, wl["data"]["Smiles"]["vocab_size"])
        m.load_oracle_state(sd)
        m.train()
        opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
        with torch.no_grad():
            x_enc, x_dec = ST.embed_inputs(sd, cfg, wl["data"], "Smiles", enc, dec)

        def one():
            opt.zero_grad(set_to_none=True)
            _, loss = m(x_enc, am, x_dec, dm, labels)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
            opt.step()
        one()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        dt2 = (time.perf_counter() - t0) / steps
        # the stock-torch wiring IS the reference's arithmetic (its layers subclass these modules) and the faster of the two:
        # it is the baseline proper; the op-by-op oracle's time is kept beside it
        res = {"value": round(cpu_batch / dt2, 4), "unit": "samples/s", "cores": cores, "host_cores": host_cores, "kind": "port",
               "sample": f"{steps} timed optimiser steps (after 1 warm-up) on a micro-batch of {cpu_batch} samples of workload {name}: "
                         f"fwd+bwd+clip+AdamW, fp32, through torch.nn.TransformerEncoder/Decoder wired as the reference wires them "
                         f"(oracle/stock_torch.py), {dt2:.2f} s/step",
               "oracle_port": port}
    return res


# ------------------------------------------------------------------------------------------------ parity, measured in this run
def measured_parity(model, wl, name, n=8):
    """Forward of `n` samples of the workload through the HIP path (eval: no dropout) and through the CPU oracle on the SAME weights
    and batch: max |logits - ref| / max |ref|, and the argmax ids (north star: 1e-3, ids equal).  The weights are the ones the
    timed steps just trained."""
    from multimodalanalytical_amd import synth
    from oracle import afm_oracle as O
    eng = model.hf_model.engine
    batch, _ = synth.make_batch(name, n, seed=4242)
    enc, am, dec, dm, labels = O.batch_to_model_inputs(batch, "Smiles")
    was = model.training
    model.eval()
    model.hf_model.backward_on_forward(False)
    with torch.no_grad():
        got = model(synth.to_device(batch, eng.dev)).logits.float().cpu().double()
    # the same batch through the TRAINING-step forward (backward pending: padded rows out of the forward pass, live positions compacted;
    # dropout is off because the model is in eval mode).  Its gradients land in the buffer the timed steps left empty: zeroed again below.
    model.hf_model.backward_on_forward(True)
    got_train = model(synth.to_device(batch, eng.dev)).logits.float().cpu().double()
    model.hf_model.backward_on_forward(False)
    eng.ps.grad.zero_()
    model.train(was)
    sd = {k: v.detach().float().cpu() for k, v in eng.state_dict().items()}
    cfg = dict(wl["cfg"], dropout=0.0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = O.model_forward(sd, cfg, wl["data"], "Smiles", enc, am, dec, dm)["logits"].double()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max()) / scale
    err_train = float((got_train - ref).abs().max()) / scale
    ids, rid = got.argmax(-1), ref.argmax(-1)
    top2 = ref.topk(2, -1).values
    sure = (top2[..., 0] - top2[..., 1]) > 2 * err * scale
    return {"logits_rel_err": float(f"{err:.3e}"), "logits_rel_err_training_step_forward": float(f"{err_train:.3e}"), "bar": 1e-3, "ids_equal_frac": round(float((ids == rid).double().mean()), 6),
            "ids_equal_where_margin_exceeds_2x_err": bool(torch.equal(ids[sure], rid[sure])),
            "decidable_frac": round(float(sure.double().mean()), 6), "samples": n,
            "note": "measured in this run on the trained weights of the timed steps, eval forward vs oracle/afm_oracle.py (fp32 CPU)"}


# ------------------------------------------------------------------------------------------------ eval / decode
def eval_decode(model, wl, name, n_beams, max_length, B, check_n=8):
    """The reference's evaluation path on the trained weights of the timed steps (modeling/wrapper.py:409-453: encoder once, then
    greedy -- every validation batch, :500-507 -- or beam search with num_return_sequences = n_beams, forced EOS at max_length),
    as this package runs it: KV cache + one HIP graph per position (greedy), afm_beam_step / afm_cache_reorder on the device (beam).
    Timed on `B` samples; ids checked in the run on `check_n` samples: greedy against the CPU oracle's full-prefix loop
    (oracle/afm_oracle.py greedy_decode = the reference's use_cache=False control flow), beam device kernels against the host loop
    that is pinned to transformers' generate (tests/golden/beam_cases.npz)."""
    from multimodalanalytical_amd import synth
    from oracle import afm_oracle as O
    eng = model.hf_model.engine
    dev = eng.dev
    old_len = model.max_length
    model.max_length = max_length
    batch = synth.make_batch(name, B, seed=777, device=dev)[0]
    kw = dict(n_beams=n_beams)
    try:
        model.generate(batch, **kw)                                   # warm-up (graph capture, allocator)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids = model.generate(batch, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ntok = ids.shape[1] - 1                                       # positions decoded (the first column is BOS)
        out = {"what": f"{'greedy' if n_beams == 1 else f'beam {n_beams}'} decode, workload {name}, B = {B}, max_length {max_length}",
               "path": "KV cache, one HIP graph per position" if n_beams == 1 else "KV cache, beam bookkeeping on the device (afm_beam_step, afm_cache_reorder)",
               "ms_per_token": round(dt / max(ntok, 1) * 1e3, 4), "tokens": ntok, "samples_per_s": round(B / dt, 2), "ms_total": round(dt * 1e3, 2),
               "returned_sequences": int(ids.shape[0])}
        small, _ = synth.make_batch(name, check_n, seed=778)
        got = model.generate(synth.to_device(small, dev), **kw).cpu()
        if n_beams == 1:
            enc, am, dec, dm, labels = O.batch_to_model_inputs(small, "Smiles")
            sd = {k: v.detach().float().cpu() for k, v in eng.state_dict().items()}
            cfg = dict(wl["cfg"], dropout=0.0)
            torch.set_num_threads(min(32, os.cpu_count() or 1))
            t0 = time.perf_counter()
            ref = O.greedy_decode(sd, cfg, wl["data"], "Smiles", enc, am, max_length=max_length)
            cpu_s = time.perf_counter() - t0
            L = min(got.shape[1], ref.shape[1])
            eq = (got[:, :L] == ref[:, :L])
            same_len = got.shape[1] == ref.shape[1]
            first = [int((~eq[i]).nonzero()[0]) if not bool(eq[i].all()) else -1 for i in range(check_n)]
            # a sequence that leaves the reference's does so at a position where the reference's own top-2 margin is inside the error band
            in_band = True
            for i, pos in enumerate(first):
                if pos < 0:
                    continue
                one = {m: ({k: t[i:i + 1] for k, t in v.items()} if isinstance(v, dict) else v[i:i + 1]) for m, v in enc.items()}
                o = O.model_forward(sd, cfg, wl["data"], "Smiles", one, am[i:i + 1], ref[i:i + 1, :pos], None)
                lg = o["logits"][0, -1].double()
                top2 = lg.topk(2).values
                in_band &= bool((top2[0] - top2[1]) <= 2e-3 * float(lg.abs().max()))
            out["ids_check"] = {"against": "oracle/afm_oracle.py greedy_decode (fp32 CPU, full-prefix recompute: the reference's use_cache=False loop)",
                                "samples": check_n, "sequences_equal": int(sum(p < 0 for p in first)), "same_length": bool(same_len),
                                "tokens_equal_frac": round(float(eq.double().mean()), 6), "first_divergence": first,
                                "divergences_inside_the_margin_band": in_band, "oracle_cpu_s": round(cpu_s, 1)}
        else:
            host = model.generate(synth.to_device(small, dev), n_beams=n_beams, device_beam=False).cpu()
            out["ids_check"] = {"against": "the host beam loop (beam.beam_search, pinned to transformers generate by tests/golden/beam_cases.npz)",
                                "samples": check_n, "sequences_equal": bool(got.shape == host.shape and torch.equal(got, host)),
                                "stop_rule": model.beam_stop_rule}
        return out
    finally:
        model.max_length = old_len


# ------------------------------------------------------------------------------------------------ timed loop
def build(workload, mode, steps_total, world, dev, args):
    from multimodalanalytical_amd import synth
    from multimodalanalytical_amd.modeling.wrapper import HFWrapper, SimpleTokenizerInfo
    from multimodalanalytical_amd.trainer import TrainLoop
    wl = synth.WORKLOADS[workload]
    tok = SimpleTokenizerInfo(wl["data"]["Smiles"]["vocab_size"])
    model = HFWrapper(wl["data"], "CustomModel", "facebook/bart-base", tok, optimiser="adamw", lr=1e-4,
                      num_steps=steps_total + 1, world_size=world, device=dev, compute_dtype=compute_dtype(mode),
                      backward_dtype=backward_dtype(mode),
                      **{k: v for k, v in wl["cfg"].items() if k != "multimodal_norm"})
    loop = TrainLoop(model, acc_batches=args.acc, world_size=world, force_reducer=args.force_ddp)
    return wl, tok, model, loop


def make_loader(workload, B, rank, world, dev, n):
    """The input path of the training loop over the resident synthetic set (reference data/datamodules.py:140-228 +
    preprocessing/patches.py:54-107 + datasets.py:58-141 for the mixture workload): rank-strided, shuffled, collated on the device."""
    from multimodalanalytical_amd import synth
    from multimodalanalytical_amd.cli.training import MixtureLoader, ShardLoader
    from multimodalanalytical_amd.preprocess import DeviceCollator, PatchPreprocessor
    wl = synth.WORKLOADS[workload]
    shard = synth.make_shard(workload, n, seed=3247, device=dev)
    if any(c.get("alignment") for c in wl["data"].values()):       # c5: mixtures of two compounds + the alignment target
        base = synth.shard_collator(workload, shard, dev)
        pre = dict(base.preprocessors)
        for m, c in wl["data"].items():
            if c.get("alignment"):
                pre[m] = PatchPreprocessor(int(c["preprocessor_arguments"]["patch_size"]), False, False, device=str(dev))
        mix = {"balanced": dict(n_compounds=2, compounds_ratio=None, train_max_n_samples=320000000, validation_max_n_samples=10000,
                                test_max_n_samples=10000, parallel_samples=16384, normalize=False)}    # configs/mixture/ir/binary.yaml
        return MixtureLoader(shard, mix, "train", DeviceCollator(wl["data"], pre), B, dev, rank, world)
    return ShardLoader(shard, synth.shard_collator(workload, shard, dev), B, dev, rank, world, shuffle=True)


def timed_run(workload, mode, steps, warmup, rank, world, dev, args, keep=False, input_path=None):
    """W untimed + K timed optimiser steps; returns (samples/s over all ranks, seconds, final loss, extras)."""
    import torch.distributed as dist
    from multimodalanalytical_amd import synth
    wl, tok, model, loop = build(workload, mode, steps + warmup, world, dev, args)
    B = args.batch or wl["batch"]
    input_path = input_path or args.input_path
    if input_path == "loader":
        loader = make_loader(workload, B, rank, world, dev, args.set_size)

        def stream():
            e = 0
            while True:
                got = False
                for b in loader.epoch(e):
                    got = True
                    yield b
                if not got:
                    raise RuntimeError("the loader produced no batch")
                e += 1
        it = stream()
        batches = [next(it) for _ in range(args.acc)]          # (shape / mask statistics only; the timed loop keeps drawing)

        def step():
            for _ in range(args.acc):
                loss = loop.micro_batch(next(it))
            return loss
    else:
        batches = [synth.make_batch(workload, B, seed=3247 + 1000 * rank + i, device=dev)[0] for i in range(args.acc)]

        def step():
            for i in range(args.acc):
                loss = loop.micro_batch(batches[i])
            return loss
    torch.cuda.synchronize()

    from multimodalanalytical_amd.trainer import barrier
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        barrier()                     # through the C-ABI communicator while it is live (trainer.barrier), else torch.distributed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        torch.cuda.synchronize()      # nothing of this rank is in flight on the side stream when torch's communicator is used
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    S = batches[0]["encoder_pad_mask"].shape[0]
    flops = synth.train_flops_per_sample(wl["cfg"], S, wl["T"], tok.vocab_size)
    live = torch.cat([(~b["encoder_pad_mask"]).sum(0).double() for b in batches])          # live encoder positions per sample
    eng_ = model.hf_model.engine
    ex = synth.executed_flops_per_sample(wl["cfg"], S, wl["T"], tok.vocab_size, float(live.mean()), float((live * live).mean()),
                                         fused_cross=bool(getattr(eng_, "xattn_fused", False)) and wl["T"] <= 128 and eng_.bd in (torch.float16, torch.bfloat16))
    res = {"value": steps * args.acc * B * world / dt, "dt": dt, "loss": float(loss), "S": S, "B": B, "flops": flops, "wl": wl,
           "executed": ex, "live_frac": float(live.mean()) / S, "input_path": input_path}
    if rank == 0 and world == 1 and not args.no_parity and not args.no_cpu_baseline:
        res["parity"] = measured_parity(model, wl, workload)
    if loop.reducer is not None:
        res["rccl_ranks"] = loop.reducer.comm.world if loop.reducer.comm is not None else 0
    if loop.reducer is not None and loop.reducer.comm is not None:      # the C-ABI RCCL communicator of this run
        torch.cuda.synchronize()
        loop.reducer.comm.close()
    if keep:
        res["model"] = model
    else:
        del model, loop, batches
    if input_path == "loader":
        del it, loader
    torch.cuda.empty_cache()
    return res


def launch_plan(gpus, force_ddp, env, argv, visible_gpus=None):
    """What this process is (reference trainer/trainer.py:58-71, cli/training.py:47-58: Lightning starts the
    ranks itself).  Pure: decides from --gpus, the launcher environment and the device count, touches nothing.
      ("rank", rank, world, local)   one rank of a job (RANK set by torch.distributed.run), or the lone process of a 1-GPU run
      ("spawn", argv)                no launcher environment and more than one rank wanted (or --force-ddp): the command of the child job
      ("error", message)             WORLD_SIZE != --gpus (always an error: a 1-rank run must never report itself as N GPUs), bad counts"""
    if gpus < 1:
        return ("error", f"--gpus {gpus}")
    if "RANK" in env:
        world = int(env.get("WORLD_SIZE", "1"))
        if world != gpus:
            return ("error", f"--gpus {gpus} but WORLD_SIZE={world}")
        rank, local = int(env["RANK"]), int(env.get("LOCAL_RANK", env["RANK"]))
        if not (0 <= rank < world):
            return ("error", f"RANK={rank} outside WORLD_SIZE={world}")
        return ("rank", rank, world, local)
    if "WORLD_SIZE" in env and int(env["WORLD_SIZE"]) != gpus:
        return ("error", f"--gpus {gpus} but WORLD_SIZE={env['WORLD_SIZE']} (and no RANK)")
    if gpus == 1 and not force_ddp:
        return ("rank", 0, 1, 0)
    if visible_gpus is not None and visible_gpus < gpus:
        return ("error", f"--gpus {gpus} but {visible_gpus} device(s) visible")
    import socket
    with socket.socket() as sk:                     # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return ("spawn", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
                      "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv))


def visible_gpu_count(env=os.environ, kfd_nodes="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process's children would see, WITHOUT loading a GPU runtime (torch.cuda.device_count() may fall back to
    hipGetDeviceCount, which brings up HIP / HSA in the parent and keeps /dev/kfd open beside the ranks): the visibility lists of the
    environment if set, else the KFD topology (nodes with SIMDs are GPUs).  None when neither is readable: the ranks then find out."""
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if env.get(k) is not None:
            return len([t for t in env[k].split(",") if t.strip() != ""])
    try:
        n = 0
        for node in os.listdir(kfd_nodes):
            with open(os.path.join(kfd_nodes, node, "properties")) as fh:
                props = dict(ln.split()[:2] for ln in fh if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        return n
    except OSError:
        return None


def spawn_job(cmd):
    """Run the N-rank job as a child process (never exec: this process may not be replaced once a GPU runtime is loaded, and must not
    touch the GPU before the children do), relay its stderr as it comes and its LAST JSON line on stdout, return its exit status."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout:
        t = ln.strip()
        if t.startswith("{") and t.endswith("}"):
            try:
                json.loads(t)
                line = t
                continue
            except ValueError:
                pass
        sys.stderr.write(ln)                         # banners of the launcher / RCCL: not part of the contract line
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 3                                       # a job that printed no JSON line did not measure anything
    return rc


def main():
    args = parse()
    spawning = "RANK" not in os.environ and (args.gpus > 1 or args.force_ddp)
    plan = launch_plan(args.gpus, args.force_ddp, os.environ, sys.argv[1:], visible_gpu_count() if spawning else None)   # (no GPU runtime in a parent)
    if plan[0] == "error":
        raise SystemExit("bench.py: " + plan[1])
    if plan[0] == "spawn":
        raise SystemExit(spawn_job(plan[1]))
    _, rank, world, local = plan
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    ddp = world > 1 or args.force_ddp
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    main_run = timed_run(args.workload, args.dtype, args.steps, args.warmup, rank, world, dev, args, keep=True)
    wl, B, S, flops = main_run["wl"], main_run["B"], main_run["S"], main_run["flops"]
    value = main_run["value"]
    # executed MFMA passes per algorithmic product: forward third of the FLOPs at 3, backward two thirds at 1 in mixed mode
    PASSES = {"fp16": 1.0, "bf16x3": 3.0, "bf16x3-mixed": (1.0 * 3 + 2.0 * 1) / 3, "bf16": 1.0, "fp32": 1.0}
    passes = PASSES[args.dtype]

    def mode_entry(r, mode, steps):
        p = PASSES[mode]
        rate = r["value"] / world / (PEAK_BF16_TFLOPS * 1e12)
        e = {"value": round(r["value"], 3), "unit": "samples/s", "ms_per_step": round(r["dt"] / steps * 1e3, 3), "steps": steps,
             "step_flop_frac": round(rate * r["flops"], 4),
             # executed: MFMA passes of the mode x (algorithmic products + the 3 products per attention instance the backward recomputes)
             "step_mfma_frac_executed": round(p * rate * r["executed"]["executed"], 4) if mode != "fp32" else None,
             "final_loss": round(r["loss"], 4)}
        if "parity" in r:
            e["logits_vs_cpu_reference"] = r["parity"]
        return e

    modes = {args.dtype: mode_entry(main_run, args.dtype, args.steps)}
    for m in [x for x in args.other_modes.split(",") if x and x != args.dtype and world == 1]:      # (N > 1: the headline mode only)
        r = timed_run(args.workload, m, args.other_steps, 1, rank, world, dev, args)
        modes[m] = mode_entry(r, m, args.other_steps)
    workloads = {}
    eval_out = {}
    if world == 1:
        for w in [x for x in args.extra_workloads.split(",") if x and x != args.workload]:
            nst = 5 if w in ("c4", "c5") else args.extra_steps
            r = timed_run(w, args.dtype, nst, 1, rank, world, dev, args, keep=w in ("c4", "c5"))
            workloads[w] = {"workload": f"{w}: modalities {'+'.join(k for k in r['wl']['data'] if k != 'Smiles')}, enc_len {r['S']}, dec_len {r['wl']['T']}, "
                                        f"{r['wl']['cfg']['encoder_layers']}L d{r['wl']['cfg']['d_model']}" + (" gated" if r['wl']['cfg']['gated_linear'] else ""),
                            "value": round(r["value"], 3), "unit": "samples/s", "ms_per_step": round(r["dt"] / nst * 1e3, 3),
                            "steps": nst,
                            "train_gflop_per_sample": round(r["flops"] / 1e9, 2),
                            # padded workloads: the algorithmic count includes positions whose work the kernels skip; `live` is the work at
                            # live encoder positions only (what a step cannot leave out), `live_frac` the share of real encoder positions
                            "live_gflop_per_sample": round(r["executed"]["live"] / 1e9, 2),
                            "encoder_live_frac": round(r["live_frac"], 4),
                            "mfma_frac_of_live_work": round(PASSES[args.dtype] * r["value"] * r["executed"]["live"] / (PEAK_BF16_TFLOPS * 1e12), 4),
                            "dtype": args.dtype}
            if "parity" in r:
                workloads[w]["logits_vs_cpu_reference"] = r["parity"]
            wsp = next((q for q in (os.path.join(ROOT, "profiles", f"{rnd}_{w}_{args.dtype}_step_pmc.json") for rnd in ("r06", "r05"))
                        if os.path.exists(q)), "")       # committed counter passes of this workload's step, newest round first
            if wsp:
                wp = json.load(open(wsp))
                workloads[w]["step_counters"] = {"source": os.path.basename(wsp), **{k: wp[k] for k in
                                                 ("mfma_busy_frac", "hbm_gbs", "clock_ghz", "hbm_bytes_per_step", "kernel_ms_per_step") if k in wp}}
            if w == "c4" and not args.no_roofline:
                # the configuration the 8-GPU target is quoted on: live HIP-event timings of ITS dominant launches
                workloads[w]["roofline_kernels"] = kernel_rooflines(r["model"], r["wl"], r["B"], args.dtype, w)
                workloads[w]["roofline_note"] = ("each entry times ONE launch over all B*S rows with every row live: the kernel's rate.  In the timed step of this "
                                                 "padded workload the encoder-row launches run over the packed live rows only (slots of ceil32(live) rows: "
                                                 f"about {min(1.0, r['live_frac'] + 16.0 / r['S']):.2f} of B*S), so their time per micro-batch is that share of ms_per_micro_batch")
            if w == "c5" and not args.no_eval:
                eval_out["beam5_c5"] = eval_decode(r["model"], r["wl"], "c5", 5, 256, r["B"])
            if "model" in r:
                del r["model"]
                torch.cuda.empty_cache()

    # the same loop over 4 fixed pre-collated batches (rounds 1-3): what the input path costs inside the timed region
    input_cmp = None
    if world == 1 and args.input_path == "loader" and not args.no_input_compare:
        r = timed_run(args.workload, args.dtype, 4, 1, rank, world, dev, args, input_path="fixed")
        input_cmp = {"timed": "loader: ShardLoader -> DeviceCollator -> afm_patch_preprocess over the resident synthetic set, inside the timed region",
                     "set_size": args.set_size, "fixed_batch_value": round(r["value"], 3), "fixed_batch_ms_per_step": round(r["dt"] / 4 * 1e3, 3),
                     "loader_over_fixed": round(value / r["value"], 4)}

    # the same global batch drawn as HALF as many micro-batches of 2 B: what the decoder side's small launches cost
    # at the reference's micro-batch.  Reported beside the headline, never as it (the benchmark's configuration is B x accumulate as given).
    mb_alt = None
    if world == 1 and args.acc % 2 == 0 and args.acc >= 2 and not args.no_input_compare:
        import copy
        a2 = copy.copy(args)
        a2.batch, a2.acc = 2 * B, args.acc // 2
        r = timed_run(args.workload, args.dtype, 4, 1, rank, world, dev, a2)
        mb_alt = {"what": f"same global batch ({B * args.acc}) as {a2.acc} micro-batches of {a2.batch}: NOT the benchmark's configuration, for comparison only",
                  "micro_batch_per_gpu": a2.batch, "acc_batches": a2.acc, "value": round(r["value"], 3), "unit": "samples/s",
                  "ms_per_step": round(r["dt"] / 4 * 1e3, 3), "over_headline": round(r["value"] / value, 4)}

    # step-level counters (committed rocprofv3 PMC passes of this command, tools/prof_step_pmc.sh): MFMA-busy share and HBM rate of a step
    step_pmc = None
    for rnd in ("r06", "r05", "r04"):
        sp = os.path.join(ROOT, "profiles", f"{rnd}_{args.workload}_{args.dtype}_step_pmc.json")
        if os.path.exists(sp):
            step_pmc = json.load(open(sp))
            break

    out = {
        "metric": METRIC, "value": round(value, 3), "unit": "samples/s",
        "n_gpus": main_run.get("rccl_ranks") or world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(main_run["dt"] / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['cfg']['encoder_layers']}L d{wl['cfg']['d_model']} "
                               f"f{wl['cfg']['encoder_ffn_dim']} enc_len {S} dec_len {wl['T']} "
                               f"modalities {'+'.join(k for k in wl['data'] if k != 'Smiles')}",
                   "workload_note": "BASELINE.json configs[1], the designated 1-GPU configuration (IR-only, S = 1024); the IR+NMR "
                                    "configuration configs[2] (c3, same 196 GFLOP/sample) is timed under `workloads`",
                   "input_path": (f"synth-{args.set_size // 1000}K ({args.set_size} samples, seed 3247) resident in HBM; every micro-batch drawn, "
                                  "standardised, patchified and collated on the device INSIDE the timed region") if args.input_path == "loader"
                   else "4 pre-collated batches per rank, cycled",
                   "micro_batch_per_gpu": B, "acc_batches": args.acc, "global_batch": B * args.acc * world,
                   "parallelism": f"dp{world}", "rccl_ranks": main_run.get("rccl_ranks", 0), "dropout": wl["cfg"]["dropout"],
                   "optimiser": "adamw+onecycle, clip 1.0",
                   "precision": {"bf16x3": "split bf16 operand pairs, 3 bf16 MFMA passes per product, fp32 accumulate / residual stream / statistics",
                                 "bf16x3-mixed": "forward as bf16x3 (split pairs, 3 MFMA passes: parity-grade logits); backward on the single-pass bf16 "
                                                 "kernels reading the hi planes of the saved pair tensors (the reference trains in 16-bit mixed precision)",
                                 "fp16": "fp16 operands forward and backward, 1 MFMA pass per product, fp32 accumulate / residual stream / statistics / "
                                         "master weights, dynamic loss scaling on the device: the reference's own GPU precision (Lightning 16-mixed)",
                                 "bf16": "bf16 operands, 1 MFMA pass, fp32 accumulate / residual stream / statistics",
                                 "fp32": "exact fp32 FMA kernels"}[args.dtype]},
        "train_gflop_per_sample": round(flops / 1e9, 2),
        "step_mfma_frac": round(value / world * flops / (PEAK_BF16_TFLOPS * 1e12), 4),
        "step_mfma_frac_executed": round(passes * value / world * main_run["executed"]["executed"] / (PEAK_BF16_TFLOPS * 1e12), 4),
        "executed_gflop_per_sample": round(main_run["executed"]["executed"] / 1e9, 2),
        "step_mfma_busy": None if step_pmc is None else step_pmc.get("mfma_busy_frac"),
        "step_hbm_gbs": None if step_pmc is None else step_pmc.get("hbm_gbs"),
        "step_counters": None if step_pmc is None else {"source": os.path.basename(sp), **{k: step_pmc[k] for k in
                                                        ("clock_ghz", "hbm_bytes_per_step", "kernel_ms_per_step") if k in step_pmc}},
        "final_loss": round(main_run["loss"], 4),
        "modes": modes,
    }
    # the driver's record keeps `config` whole and only the NAMES of the nested sections: the other workloads' rates and the headline
    # mode's in-run parity go there as short scalars too
    for w, e in workloads.items():
        out["config"][f"{w}_samples_per_s"] = e["value"]
    par = main_run.get("parity")
    parity_ok = None
    if par is not None:
        worst = max(par["logits_rel_err"], par["logits_rel_err_training_step_forward"])
        parity_ok = bool(worst < par["bar"] and par["ids_equal_where_margin_exceeds_2x_err"])
        out["config"].update(logits_rel_err=par["logits_rel_err"], logits_rel_err_training_step_forward=par["logits_rel_err_training_step_forward"],
                             logits_bar=par["bar"], ids_equal_frac=par["ids_equal_frac"])
        for w, e in workloads.items():
            if "logits_vs_cpu_reference" in e:
                pw = e["logits_vs_cpu_reference"]
                out["config"][f"{w}_logits_rel_err"] = max(pw["logits_rel_err"], pw["logits_rel_err_training_step_forward"])
                parity_ok = parity_ok and bool(out["config"][f"{w}_logits_rel_err"] < pw["bar"] and pw["ids_equal_where_margin_exceeds_2x_err"])
    out["parity_ok"] = parity_ok      # None: the in-run check was switched off (--no-parity / --no-cpu-baseline)
    if input_cmp is not None:
        out["input_path"] = input_cmp
    if mb_alt is not None:
        out["micro_batch_alt"] = mb_alt
    if workloads:
        out["workloads"] = workloads
    if rank == 0:
        if world == 1 and not args.no_eval:
            eval_out["greedy_c2" if args.workload == "c2" else f"greedy_{args.workload}"] = eval_decode(main_run["model"], wl, args.workload, 1, 128, B)
        if eval_out:
            out["eval"] = eval_out
        if not args.no_roofline:
            ks = kernel_rooflines(main_run["model"], wl, B, args.dtype, args.workload)
            out["roofline"] = ks[0]
            out["roofline_kernels"] = ks[1:]
            # share of the timed step the listed launches account for (their live timings x launches per step / the step's wall time)
            per_step = sum(e["ms_per_micro_batch"] * args.acc + e["avg_launch_ms"] * e.get("launches_per_step", 0) for e in ks)
            out["roofline_coverage_of_step"] = round(per_step / (main_run["dt"] / args.steps * 1e3), 4)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(main_run["model"], wl, args.workload, args.cpu_batch, args.cpu_steps, args.cpu_threads)
    # RCCL writes its version banner through C stdio, which holds it (stdout is a pipe) until the process exits -- behind the JSON
    # line.  Every rank flushes its C buffers BEFORE the last barrier, rank 0 prints after it: the JSON line is the last line of the
    # job's stdout, whatever the launcher.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if ddp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if out.get("parity_ok") is False:      # the line is printed first; a headline whose in-run logits miss the bar exits with status 4
        sys.stderr.write("bench.py: the in-run logits / ids check against the CPU oracle missed its bar (parity_ok: false)\n")
        raise SystemExit(4)


if __name__ == "__main__":
    main()
