"""Fuzz of the fused short-query attention backward: random shapes, masks, dropout and dtypes; every case (a) twice, bit for bit (no atomics
in the kernel: any difference is a race), (b) against the two general kernels within rounding.  `python tools/experiments/fsq_fuzz.py [cases] [seed]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops

DEV = "cuda:0"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
for case in range(n_cases):
    B, H, dh = ri(1, 9), ri(1, 9), 64
    Tq = ri(1, 128)
    Tk = 2 * ri(1, 700)
    p = 0.1 if ri(0, 1) else 0.0
    dtype = torch.float16 if ri(0, 1) else torch.bfloat16
    packed = ri(0, 3) == 0 and Tk % 128 == 0
    if ri(0, 3) == 0:
        Tk = 128 * ri(1, 10)
        packed = ri(0, 1) == 1
    causal = ri(0, 3) == 0          # the decoder's self-attention: Tq == Tk <= 128, dense rows
    if causal:
        Tq = 2 * ri(1, 64); Tk = Tq; packed = False
    d = H * dh
    n = torch.randint(0, Tk + 1, (B,), generator=g)
    if ri(0, 2) == 0:
        n[ri(0, B - 1)] = Tk
    pad = (torch.arange(Tk)[None, :] >= n[:, None]).to(torch.uint8).to(DEV)
    q = (torch.randn(B * Tq, d, generator=g) * 0.5).to(DEV).to(dtype)
    kv_d = (torch.randn(B * Tk, 2 * d, generator=g) * 0.5).to(DEV).to(dtype)
    do = (torch.randn(B * Tq, d, generator=g) * 0.1).to(DEV).to(dtype)
    kw = {}
    kv = kv_d
    dest = None
    if packed:
        plan = ops.compact_plan(pad, B, Tk, 256, compact=2) if Tk % 256 == 0 else None
        if plan is None:
            packed = False
        else:
            dest = plan.dest.long()
            kv = torch.zeros_like(kv_d); kv[dest] = kv_d
            kw = dict(k_off=plan.seq_off)
    drop = ops.drop(p, 4 + case, 1) if p else ops.NO_DROP

    def run(flag, dense=False):
        kv, kw = (kv_d, {}) if dense else (kv_run, kw_run)
        shp = ops.attn_shape(B, H, Tq, Tk, dh, dtype, d, 2 * d, 2 * d, d, pad, causal, drop, **kw)
        if p:
            ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=DEV))
        o = torch.full((B * Tq, d), 3.0, dtype=dtype, device=DEV)
        lse = torch.full((B * H * Tq,), 3.0, device=DEV)
        ops.attn_fwd(shp, q, kv[:, :d], kv[:, d:], o, lse)
        shp.reserved |= flag
        dq = torch.full((B * Tq, d), 3.0, dtype=dtype, device=DEV)
        dkv = torch.full((B * Tk, 2 * d), 3.0, dtype=dtype, device=DEV)
        delta = torch.full_like(lse, 7.0)
        ops.attn_bwd(shp, q, kv[:, :d], kv[:, d:], o, do, lse, delta, dq, dkv[:, :d], dkv[:, d:], d, 2 * d, 2 * d)
        return ops.last_algo(), dq, dkv, delta

    kv_run, kw_run = kv, kw
    a0, dq0, dkv0, dl0 = run(32768, dense=True)      # the two general kernels, always on the dense layout
    a1, dq1, dkv1, dl1 = run(262144)
    a2, dq2, dkv2, dl2 = run(262144)
    msg = []
    if a1 != "attn_fsq":
        msg.append(f"not taken ({a1})")
    else:
        if not (torch.equal(dq1, dq2) and torch.equal(dkv1, dkv2) and torch.equal(dl1, dl2)):
            msg.append("NOT DETERMINISTIC")
        if not bool(torch.isfinite(dkv1.float()).all() and torch.isfinite(dq1.float()).all()):
            msg.append("non-finite")
        tol = 4e-3 if dtype == torch.float16 else 3e-2
        one_hot = (bool((n == 1).any()) or causal) and p > 0          # (see tests/test_gpu_attn_fsq.py: the rounded-output noise of one-hot softmax rows)
        if packed:      # packed -> dense order; the dead tail must hold zeros
            used = int(plan.seq_off[-1])
            if float(dkv1[used:].float().abs().max() if used < B * Tk else 0.0) != 0.0:
                msg.append("dead tail not zero")
            live = (pad == 0).view(-1)
            back = torch.zeros_like(dkv1); back[live] = dkv1[dest][live]
            dkv1 = back
            dkv0 = dkv0.clone(); dkv0[~live] = 0
        for nm, x, y in (("dQ", dq0, dq1), ("dK", dkv0[:, :d], dkv1[:, :d]), ("dV", dkv0[:, d:], dkv1[:, d:])):
            e = float((x.float() - y.float()).abs().max() / x.float().abs().max().clamp_min(5e-3))      # (floor: where every row is one-hot the exact gradient is 0 and both kernels return noise)
            if e > (10 * tol if one_hot else tol):
                msg.append(f"{nm} {e:.2e}")
        if not torch.equal(dl0, dl1):
            msg.append("delta differs")
    if msg:
        bad += 1
        print(f"case {case}: B{B} H{H} Tq{Tq} Tk{Tk} p{p} {dtype} packed={packed} causal={causal} lens={n.tolist()}: {'; '.join(msg)}", flush=True)
print(f"{n_cases} cases, {bad} with findings", flush=True)
