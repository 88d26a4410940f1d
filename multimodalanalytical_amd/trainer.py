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


_NATIVE = None      # (NativeComm, side stream) of the live BucketedReducer, if any


def sync_mean(value, group=None):
    """Lightning's `self.log(..., sync_dist=True)` (reference wrapper.py:474,486,601): the logged scalar is averaged
    over the ranks.  Without an initialised process group (single GPU) the value is returned unchanged.  While a reducer with
    the C-ABI RCCL communicator is live, device scalars travel through THAT communicator on its side stream: the logged loss is
    reduced while gradient buckets are in flight, and collectives of two communicators issued from two streams have no defined
    relative order across ranks (one communicator, one stream: issue order = execution order everywhere)."""
    if not (dist.is_available() and dist.is_initialized()):
        return value
    world = dist.get_world_size(group)
    if world == 1:
        return value
    if _NATIVE is not None and torch.is_tensor(value) and value.is_cuda and group is None:
        return _native_mean(value, *_NATIVE)
    t = value.detach().clone().float() if torch.is_tensor(value) else torch.tensor(float(value))
    if t.device.type == "cpu" and dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t / world


def sync_flag(flag: bool, src: int = 0, device=None, group=None) -> bool:
    """Rank `src`'s boolean, on every rank (loop control that only one rank can decide: early stopping follows the checkpoint
    ranking kept on rank 0; every rank must leave the epoch loop in the same epoch or the others hang in the next collective)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(flag)
    if _NATIVE is not None and group is None:
        # ONE communicator, ONE stream while the C-ABI exchange is live (see sync_mean): the flag is a sum to which only `src` contributes
        comm, side = _NATIVE
        t = torch.tensor([1.0 if (flag and comm.rank == src) else 0.0], device=device or torch.device("cuda", torch.cuda.current_device()))
        return bool(float(_native_mean(t, comm, side).item()) > 0.0)
    dev = device if (device is not None and dist.get_backend(group) == "nccl") else "cpu"
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.broadcast(t, src=src, group=group)
    return bool(int(t.item()))


def barrier(group=None) -> None:
    """All ranks have reached this point.  While the C-ABI communicator is live it carries the barrier too (a one-float all-reduce on
    the side stream, then a host wait): a torch-communicator collective enqueued while a native bucket is still in flight has no
    defined order against it across ranks.  Otherwise torch.distributed.barrier."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    if _NATIVE is not None and group is None:
        comm, side = _NATIVE
        t = torch.zeros(1, device=torch.device("cuda", torch.cuda.current_device()))
        _native_mean(t, comm, side)
        torch.cuda.current_stream().synchronize()
        return
    if torch.cuda.is_available() and dist.get_backend(group) == "nccl":
        torch.cuda.synchronize()        # nothing of this process is still in flight on another stream when the collective is enqueued
    dist.barrier(group=group)


def _native_mean(value: torch.Tensor, comm, side) -> torch.Tensor:
    """Mean over the ranks of a device scalar through the C-ABI communicator, on its side stream."""
    t = value.detach().float().reshape(-1).clone()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    side.wait_event(ev)
    comm.all_reduce(t, side)
    t.record_stream(side)
    torch.cuda.current_stream().wait_stream(side)
    return (t / comm.world).reshape(value.shape)


class NativeComm:
    """RCCL communicator behind the C ABI (include/afm_hip.h: afm_comm_create / afm_allreduce_bucket).  torch.distributed is
    only the bootstrap: rank 0's 128-byte unique id is broadcast over the existing process group."""

    def __init__(self, group=None):
        import ctypes as C
        from . import lib as L
        self._L, self._C = L, C
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = (C.c_ubyte * 128)()
        status = 0
        if rank == 0:       # rank 0 ALWAYS reaches the broadcast, with a failure marker if it has no id (raising here
            try:            # left the other ranks in the broadcast while rank 0 went on to the MIN agreement)
                status = int(L.load().afm_comm_unique_id(buf))
            except Exception:      # noqa: BLE001
                status = -100
        ids = [(status, bytes(buf))]
        dist.broadcast_object_list(ids, src=0, group=group)
        status, payload = ids[0]
        self.handle = C.c_void_p()
        if status != 0:
            raise L.AfmError(f"afm_comm_unique_id failed on rank 0 ({status})")
        raw = (C.c_ubyte * 128).from_buffer_copy(payload)
        L.check(L.load().afm_comm_create(C.byref(self.handle), raw, rank, world), "afm_comm_create")
        r, w = C.c_int32(-1), C.c_int32(-1)
        L.check(L.load().afm_comm_count(self.handle, C.byref(r), C.byref(w)), "afm_comm_count")
        self.rank, self.world = int(r.value), int(w.value)      # what the communicator itself reports (ncclCommUserRank / ncclCommCount)
        if (self.rank, self.world) != (rank, world):
            raise L.AfmError(f"communicator reports rank {self.rank} of {self.world}, process group {rank} of {world}")

    def all_reduce(self, view: torch.Tensor, stream) -> None:
        assert view.is_cuda and view.dtype == torch.float32 and view.is_contiguous()
        self._L.check(self._L.load().afm_allreduce_bucket(self.handle, view.data_ptr(), view.numel(), stream.cuda_stream),
                      "afm_allreduce_bucket")

    def close(self) -> None:
        global _NATIVE
        if _NATIVE is not None and _NATIVE[0] is self:
            _NATIVE = None
        if self.handle:
            self._L.load().afm_comm_destroy(self.handle)
            self.handle = None


def bucket_elems_for(numel: int, target_buckets: int = 8, floor: int = 1 << 20) -> int:
    """About eight buckets per exchange whatever the model size (c2: 47 M gradients -> 6 M floats = 24 MB per bucket, c4: 256 M ->
    32 M floats): the last bucket (embeddings + first encoder layer) is the only one that cannot overlap the backward pass,
    and xGMI ring all-reduce is bandwidth-bound from a few MB upwards."""
    return max(floor, -(-int(numel) // target_buckets))


class BucketedReducer:
    """All-reduce a flat gradient buffer back to front in buckets, on a side stream.

    `ready(lo)` declares that every gradient at flat offset >= lo is final; whole buckets below the
    previous mark are launched immediately.  `finish()` flushes the remainder and makes the compute
    stream wait for the exchange.  On HIP the buckets go through the C ABI's afm_allreduce_bucket (RCCL; `native`, the
    default; AFM_NATIVE_RCCL=0 keeps torch.distributed's nccl backend); CPU tensors use gloo (tests)."""

    def __init__(self, flat: torch.Tensor, bucket_elems: int = 0, group=None, native: Optional[bool] = None):
        self.flat, self.group = flat, group
        self.bucket = int(bucket_elems) or bucket_elems_for(flat.numel())
        self.hi = flat.numel()
        self.is_cuda = flat.is_cuda
        self.stream = torch.cuda.Stream() if self.is_cuda else None
        self.handles: List = []
        self.launched: List[Tuple[int, int]] = []
        import os as _os
        self._standin = self.is_cuda and _os.environ.get("AFM_DDP_STANDIN", "0") == "1" and (not dist.is_initialized() or dist.get_world_size(group) == 1)
        self._scratch = None
        if native is None:
            import os
            native = self.is_cuda and os.environ.get("AFM_NATIVE_RCCL", "1") != "0" and dist.is_initialized() and \
                dist.get_backend(group) == "nccl"
        self.comm = None
        if native:
            # every rank tries; the ranks then agree (MIN over the process group) so that one rank's failure -- librccl not
            # found, a communicator the fabric refuses -- moves ALL of them to torch.distributed's all-reduce instead of
            # leaving the job with two exchange paths that never meet
            comm, err = None, None
            try:
                comm = NativeComm(group)
            except Exception as e:      # noqa: BLE001
                err = e
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=flat.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 1:
                self.comm = comm
            else:
                if comm is not None:
                    comm.close()
                import warnings
                warnings.warn(f"native RCCL communicator unavailable on at least one rank ({err!r} here): "
                              "gradient buckets go through torch.distributed.all_reduce")
        if self.comm is not None and group is None:
            global _NATIVE
            _NATIVE = (self.comm, self.stream)

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
            if self.comm is not None:
                self.comm.all_reduce(view, self.stream)    # enqueued on the side stream; finish() orders the compute stream behind it
                if self._standin:
                    # profiling aid (AFM_DDP_STANDIN=1, one rank only): RCCL enqueues NOTHING for an in-place all-reduce over one rank, so a
                    # trace of `bench.py --force-ddp` on a 1-GPU box shows no exchange at all.  A device copy of the bucket on the same side
                    # stream stands in for it -- the bucket's bytes read and written once, about what a ring all-reduce moves per GPU --
                    # so the timeline shows where the buckets are launched and what runs beside them (tools/rocpd_overlap.py --standin).
                    if self._scratch is None or self._scratch.numel() < view.numel():
                        self._scratch = torch.empty(self.bucket, dtype=view.dtype, device=view.device)
                    with torch.cuda.stream(self.stream):
                        self._scratch[:view.numel()].copy_(view)
            else:
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

    def __init__(self, model, acc_batches: int = 4, world_size: int = 1, bucket_elems: int = 0,
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

    def flush(self) -> None:
        """End of an epoch with a partly filled accumulation window: step on what has been accumulated (Lightning steps on the
        last batch of every epoch; `calculate_training_steps` counts ceil(batches / acc) steps per epoch accordingly).  The
        micro-batches were scaled by 1/acc like any others, as Lightning scales them."""
        if self.micro % self.acc == 0:
            return
        if self.reducer is not None:       # no bucket was launched during these backward passes: one exchange of the whole buffer
            self.reducer.reset()
            self.reducer.finish()
        self.optim.step(grads_are_summed_over_ranks=self.reducer is not None)
        self.micro += self.acc - self.micro % self.acc


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
        if opt.get("loss_scaler") is not None:
            # fp16 mode: {S, growth tracker, steps taken, steps skipped}.  k_adam takes its bias corrections from "steps taken", so a
            # resume without it would restart them at t = 1 on warm moments
            ckpt["optimizer_states"][0]["loss_scaler"] = opt["loss_scaler"].cpu()
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
        scaler = st.get("loss_scaler")
        if scaler is None and getattr(model.hf_model.engine, "scaler", None) is not None:
            # a checkpoint written before the scaler was stored (or by a non-fp16 run): every optimiser step counted as taken, fresh scale
            scaler = model.hf_model.engine.scaler.detach().cpu().clone()
            scaler[1], scaler[2], scaler[3] = 0.0, float(st["step"]), 0.0
        loop.optim.load_state_dict({"step": st["step"], "exp_avg": st["exp_avg"].to(dev), "exp_avg_sq": st["exp_avg_sq"].to(dev),
                                    "loss_scaler": scaler})
    return ckpt
