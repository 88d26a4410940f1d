"""Training loop semantics of the reference's Lightning trainer (reference trainer/trainer.py:9-73:
accumulate_grad_batches, gradient_clip_val, DDP) and `calculate_training_steps`
(reference utils.py:156-172), without Lightning.

Data parallelism: one process per GPU, full replica each, disjoint shard of every
global batch, ONE exchange per optimiser step: all-reduce(sum) of the flat fp32 gradient buffer in
reverse-order buckets on a side HIP stream (RCCL via torch.distributed "nccl"), issued as the
backward pass finishes each layer so the exchange overlaps the rest of backward; the mean (1/world)
is folded into the fused clip+Adam kernel.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def calculate_training_steps(len_train: int, batch_size: int, acc_batches: int, epochs: int,
                             world_size: int = 1) -> int:
    """utils.py:156-172 (the reference hard-codes world_size = 1; pass 1 for its behaviour)."""
    batches_per_gpu = math.ceil((len_train / batch_size) / float(world_size))
    return math.ceil(batches_per_gpu / acc_batches) * epochs


def sync_mean(value, group=None):
    """Lightning's `self.log(..., sync_dist=True)` (reference wrapper.py:474,486,601): the logged scalar is averaged
    over the ranks.  Without an initialised process group (single GPU) the value is returned unchanged."""
    if not (dist.is_available() and dist.is_initialized()):
        return value
    world = dist.get_world_size(group)
    if world == 1:
        return value
    t = value.detach().clone().float() if torch.is_tensor(value) else torch.tensor(float(value))
    if t.device.type == "cpu" and dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t / world


class BucketedReducer:
    """All-reduce a flat gradient buffer back to front in buckets, on a side stream.

    `ready(lo)` declares that every gradient at flat offset >= lo is final; whole buckets below the
    previous mark are launched immediately.  `finish()` flushes the remainder and makes the compute
    stream wait for the exchange.  Works on CPU tensors with gloo (tests) and on HIP with RCCL."""

    def __init__(self, flat: torch.Tensor, bucket_elems: int = 16 << 20, group=None):
        self.flat, self.group = flat, group
        self.bucket = int(bucket_elems)
        self.hi = flat.numel()
        self.is_cuda = flat.is_cuda
        self.stream = torch.cuda.Stream() if self.is_cuda else None
        self.handles: List = []
        self.launched: List[Tuple[int, int]] = []

    def reset(self):
        self.hi = self.flat.numel()
        self.handles.clear()
        self.launched.clear()

    def _launch(self, lo: int, hi: int):
        if hi <= lo:
            return
        view = self.flat[lo:hi]
        self.launched.append((lo, hi))
        if self.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)                 # gradients in [lo, hi) are produced before this point
            with torch.cuda.stream(self.stream):
                self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def ready(self, lo: int):
        lo = max(0, min(int(lo), self.hi))
        while self.hi - lo >= self.bucket:
            self._launch(self.hi - self.bucket, self.hi)
            self.hi -= self.bucket

    def finish(self):
        self._launch(0, self.hi)
        self.hi = 0
        for h in self.handles:
            h.wait()                                    # on HIP: makes the current stream wait, not the host
        if self.is_cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.handles.clear()


class TrainLoop:
    """accumulate -> (all-reduce) -> clip + Adam + OneCycle, driving an HFWrapper."""

    def __init__(self, model, acc_batches: int = 4, world_size: int = 1, bucket_elems: int = 16 << 20,
                 force_reducer: bool = False):
        self.model, self.acc, self.world = model, int(acc_batches), int(world_size)
        (self.optim,), _ = model.configure_optimizers()
        self.micro = 0
        eng = model.hf_model.engine
        self.reducer = BucketedReducer(eng.ps.grad, bucket_elems) if (self.world > 1 or force_reducer) else None
        eng.grad_ready_hook = None

    def micro_batch(self, batch, batch_idx: Optional[int] = None) -> torch.Tensor:
        eng = self.model.hf_model.engine
        boundary = (self.micro + 1) % self.acc == 0
        if self.reducer is not None and boundary:      # DDP no_sync on the other micro-batches
            self.reducer.reset()
            eng.grad_ready_hook = self.reducer.ready
        loss = self.model.training_step(batch, self.micro if batch_idx is None else batch_idx, 1.0 / self.acc)
        eng.grad_ready_hook = None
        self.micro += 1
        if boundary:
            if self.reducer is not None:
                self.reducer.finish()
            self.optim.step(grads_are_summed_over_ranks=self.reducer is not None)
        return loss


def save_checkpoint(path: str, model, loop: "TrainLoop" = None, epoch: int = 0) -> None:
    """Lightning-shaped checkpoint (reference trainer/trainer.py:31-37, cli/training.py:152-183 read
    `checkpoint["state_dict"]` with `hf_model.*` / `multimodal_embedding.*` keys): the state dict uses the
    reference's key names, so `checkpoint["state_dict"]` written here loads into the reference model and vice
    versa.  The interop is the state dict only: optimiser moments are stored flat (ParamStore order), not in
    torch.optim's per-parameter layout, so Lightning's `fit(ckpt_path=)` cannot resume from these files."""
    eng = model.hf_model.engine
    ckpt = {"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
            "epoch": int(epoch), "global_step": 0 if loop is None else loop.optim.step_count,
            "pytorch-lightning_version": "2.5.1 (multimodalanalytical_amd writer)",
            "dropout_stream": {"micro_step": int(eng.micro_step), "seed": int(eng.dropout_seed)}}
    if loop is not None:
        ckpt["micro"] = int(loop.micro)
    if loop is not None:
        opt = loop.optim.state_dict()
        ckpt["optimizer_states"] = [{"step": opt["step"], "exp_avg": opt["exp_avg"].cpu(), "exp_avg_sq": opt["exp_avg_sq"].cpu(),
                                     "layout": "flat (params.ParamStore offsets)"}]
    torch.save(ckpt, path)


def load_checkpoint(path: str, model, loop: "TrainLoop" = None, strict: bool = True) -> dict:
    """Counterpart of `model.load_state_dict(checkpoint["state_dict"])` in the reference CLIs
    (cli/predict.py:114-115, cli/training.py:152-163: `align_network.*` keys are dropped when unused)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"]
    eng = model.hf_model.engine
    if eng.align is None:      # cli/training.py:152-161: only a model WITHOUT an alignment head drops the keys
        sd = {k: v for k, v in sd.items() if "align_network" not in k}
    model.load_state_dict(sd, strict=strict)
    if "dropout_stream" in ckpt:   # position of the counter-based dropout stream: a resumed run continues it
        eng.micro_step = int(ckpt["dropout_stream"]["micro_step"])
        eng.dropout_seed = int(ckpt["dropout_stream"]["seed"])
    if loop is not None and "micro" in ckpt:
        loop.micro = int(ckpt["micro"])
    if loop is not None and ckpt.get("optimizer_states") and "exp_avg" in ckpt["optimizer_states"][0]:
        st = ckpt["optimizer_states"][0]
        dev = model.hf_model.engine.dev
        loop.optim.load_state_dict({"step": st["step"], "exp_avg": st["exp_avg"].to(dev), "exp_avg_sq": st["exp_avg_sq"].to(dev)})
    return ckpt
