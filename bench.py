#!/usr/bin/env python3
"""Headline benchmark: train samples/sec of the spectra->SMILES path on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W            # N > 1: launched by torch.distributed.run

A "step" is one optimiser step of the reference's training configuration: acc_batches (4)
micro-batches of `batch` (128) samples each through HFWrapper.training_step (forward + backward,
dropout 0.1 active), then gradient clip 1.0 + AdamW + OneCycleLR; with N > 1 the flat gradient
buffer is all-reduced over RCCL once per step, overlapped with the last backward.  Inputs are
synthetic (multimodalanalytical_amd/synth.py, seeded) and resident in HBM before the timed region.
Prints ONE JSON line (rank 0).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", help="c1..c5 (multimodalanalytical_amd/synth.py)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU micro-batch (default: the workload's, 128)")
    ap.add_argument("--acc", type=int, default=4)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "bf16x3", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--force-ddp", action="store_true",
                    help="run the RCCL gradient exchange even with one rank (exercises the N>1 code path on a 1-GPU box)")
    return ap.parse_args()


def time_kernel(fn, iters=10, warm=3):
    """Average device time of fn() in ms, HIP events on the stream the kernel is launched on."""
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


def dominant_kernel_roofline(model, wl, B, dtype):
    """The weight-gradient GEMM (k_gemm_tn_ring256): with the attention dK/dV kernel (VALU-bound, DESIGN.md
    section 4) one of the two kernels with the largest share of the step in the rocprofv3 summary
    (profiles/r01_c2_kernel_stats.csv, ~14 % each), and the one with a clean roofline; timed here at its
    largest shape -- the FFN up-projection wgrad dW1[f x d] += dU^T[f x B*S] X[B*S x d] with the bias
    gradient fused (one launch = 2*B*S*d*f FLOPs).  Algorithmic bytes per launch: dU and X read once
    (bf16) + dW1 written once (fp32) = 2*B*S*(f+d) + 4*f*d; at 8 TB/s that is less time than the
    FLOPs take at the dense bf16 MFMA peak, so the MFMA roof is the binding one.
    `secondary` is the same measurement for the forward launch of that layer (x W1^T with the
    bias + GELU + dropout epilogue, keep*scale*GELU' stored for backward), whose 2 x 537 MB of output make HBM its roof."""
    from multimodalanalytical_amd import ops
    from multimodalanalytical_amd.lib import ACT_GELU_SAVE_GRAD
    cfg = wl["cfg"]
    S = sum(v[0] if isinstance(v, tuple) else v for v in wl["lens"].values())
    d, f = cfg["d_model"], cfg["encoder_ffn_dim"] * (2 if cfg["gated_linear"] else 1)
    M = B * S
    eng = model.hf_model.engine
    x = torch.randn(M, d, device=eng.dev).to(eng.cd)
    du = torch.randn(M, f, device=eng.dev).to(eng.cd)
    gw = torch.zeros(f, d, device=eng.dev)
    gb = torch.zeros(f, device=eng.dev)
    ms = time_kernel(lambda: ops.gemm(du, x, gw, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb))
    algo = ops.last_algo()
    flops = 2.0 * M * d * f
    ach = flops / (ms * 1e-3) / 1e12
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else 157.3
    esz = 2 if dtype == "bf16" else 4
    # HBM bytes per launch of this very kernel/shape from the committed PMC passes (FETCH_SIZE x2 +
    # WRITE_SIZE, profiles/r01_wgrad_ffn1_pmc.json); null when the shape differs from the profiled one
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_wgrad_ffn1_pmc.json")
    if os.path.exists(pmc) and (M, f, d) == (131072, 2048, 512) and dtype == "bf16":
        traffic = json.load(open(pmc)).get("traffic_bytes_per_launch")
    out = {"bound": "mfma", "kernel": f"afm_gemm[{algo}] wgrad dW1 {f}x{d} over {M} tokens (FFN linear1, bias gradient fused)",
           "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
           "traffic": traffic, "traffic_unit": "HBM bytes/launch (PMC)",
           "algorithmic_bytes": esz * M * (f + d) + 4 * f * d, "avg_launch_ms": round(ms, 4)}
    if not cfg["gated_linear"]:
        w = eng.W("encoder.layers.0.linear1.weight", f, d)
        o = torch.empty(M, f, dtype=eng.cd, device=eng.dev)
        bias = eng.ps.p("encoder.layers.0.linear1.bias")
        pre = torch.empty_like(o)
        dr = ops.drop(cfg["dropout"], 1, 1)
        ms2 = time_kernel(lambda: ops.gemm(x, w, o, trans_b=True, bias=bias, act=ACT_GELU_SAVE_GRAD, pre_act=pre, dropout=dr))
        by = esz * (M * d + f * d) + 2 * esz * M * f
        tr2 = None
        pmc2 = os.path.join(ROOT, "profiles", "r01_ffn1_gemm_pmc.json")
        if os.path.exists(pmc2) and (M, f, d) == (131072, 2048, 512) and dtype == "bf16":
            tr2 = json.load(open(pmc2)).get("traffic_bytes_per_launch")
        out["secondary"] = {"bound": "hbm", "kernel": f"afm_gemm[{ops.last_algo()}] {M}x{f}x{d} (FFN linear1 forward, fused bias+GELU+dropout, keep*scale*GELU' stored for backward)",
                            "achieved": round(by / (ms2 * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                            "frac": round(by / (ms2 * 1e-3) / 1e9 / 8000.0, 4), "traffic": tr2,
                            "algorithmic_bytes": by, "avg_launch_ms": round(ms2, 4),
                            "tflops": round(flops / (ms2 * 1e-3) / 1e12, 1)}
    return out


def cpu_baseline(model, wl, name, cpu_batch, threads):
    """The CPU oracle (oracle/afm_oracle.py, a port of the reference arithmetic) on this host's
    cores: forward + backward + clip + AdamW of ONE micro-batch of `cpu_batch` samples of the same
    workload, same weights."""
    from multimodalanalytical_amd import synth
    from oracle import afm_oracle as O
    cores = max(1, min(threads, os.cpu_count() or 1))   # torch CPU thread pool size actually used
    torch.set_num_threads(cores)
    eng = model.hf_model.engine
    sd = {k: v.detach().float().cpu() for k, v in eng.state_dict().items() if not k.startswith("decoder.embedding.")}
    batch, _ = synth.make_batch(name, cpu_batch, seed=99)
    enc, am, dec, dm, labels = O.batch_to_model_inputs(batch, "Smiles")
    cfg = dict(wl["cfg"]); cfg["dropout"] = 0.0
    tr = O.OracleTrainer(sd, cfg, wl["data"], "Smiles", lr=1e-4, total_steps=10, acc_batches=1)
    t0 = time.perf_counter()
    tr.micro_batch(enc, am, dec, dm, labels)
    dt = time.perf_counter() - t0
    return {"value": round(cpu_batch / dt, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"1 optimiser step on 1 micro-batch of {cpu_batch} samples of workload {name} "
                      f"(fwd+bwd+clip+AdamW, fp32, {dt:.1f} s)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    ddp = world > 1 or args.force_ddp
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    from multimodalanalytical_amd import synth
    from multimodalanalytical_amd.modeling.wrapper import HFWrapper, SimpleTokenizerInfo
    from multimodalanalytical_amd.trainer import TrainLoop

    wl = synth.WORKLOADS[args.workload]
    B = args.batch or wl["batch"]
    from multimodalanalytical_amd.x2 import X2
    cd = {"bf16": torch.bfloat16, "bf16x3": X2.dtype, "fp32": torch.float32}[args.dtype]
    tok = SimpleTokenizerInfo(wl["data"]["Smiles"]["vocab_size"])
    model = HFWrapper(wl["data"], "CustomModel", "facebook/bart-base", tok, optimiser="adamw", lr=1e-4,
                      num_steps=args.steps + args.warmup + 1, world_size=world, device=dev, compute_dtype=cd,
                      **{k: v for k, v in wl["cfg"].items() if k != "multimodal_norm"})
    loop = TrainLoop(model, acc_batches=args.acc, world_size=world, force_reducer=args.force_ddp)
    # synthetic shard of this rank, resident in HBM before timing
    batches = [synth.make_batch(args.workload, B, seed=3247 + 1000 * rank + i, device=dev)[0] for i in range(args.acc)]
    torch.cuda.synchronize()

    def step():
        for i in range(args.acc):
            loss = loop.micro_batch(batches[i])
        return loss

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss)
    samples = args.steps * args.acc * B * world
    value = samples / dt
    S = batches[0]["encoder_pad_mask"].shape[0]
    flops = synth.train_flops_per_sample(wl["cfg"], S, wl["T"], tok.vocab_size)
    out = {
        "metric": "train samples/sec (IR+NMR->SMILES, enc1024/dec128)", "value": round(value, 3), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['cfg']['encoder_layers']}L d{wl['cfg']['d_model']} "
                               f"f{wl['cfg']['encoder_ffn_dim']} enc_len {S} dec_len {wl['T']} "
                               f"modalities {'+'.join(k for k in wl['data'] if k != 'Smiles')}",
                   "micro_batch_per_gpu": B, "acc_batches": args.acc, "global_batch": B * args.acc * world,
                   "parallelism": f"dp{world}", "dropout": wl["cfg"]["dropout"], "optimiser": "adamw+onecycle, clip 1.0"},
        "train_gflop_per_sample": round(flops / 1e9, 2),
        "step_mfma_frac": round(value / world * flops / (PEAK_BF16_TFLOPS * 1e12), 4),
        "final_loss": round(loss_val, 4),
    }
    if rank == 0:
        if not args.no_roofline:
            out["roofline"] = dominant_kernel_roofline(model, wl, B, args.dtype)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, wl, args.workload, args.cpu_batch, args.cpu_threads)
        print(json.dumps(out), flush=True)
    if ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
