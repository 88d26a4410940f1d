"""Optimiser step of the training path: global-norm clip + Adam/AdamW + OneCycleLR.

Mirrors `HFWrapper.configure_optimizers` (reference modeling/wrapper.py:329-344: Adam or AdamW
over ALL parameters, OneCycleLR(max_lr=lr, total_steps=num_steps) stepped once per optimiser
step) and Lightning's `gradient_clip_val` (trainer/trainer.py:65).  The schedule is evaluated on
the host (a handful of scalars), everything that touches parameters is two kernel launches over
the flat buffers: afm_sumsq and afm_adam_step.  No host<->device synchronisation.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

from . import ops


def onecycle(step: int, total_steps: int, max_lr: float, pct_start: float = 0.3, div_factor: float = 25.0,
             final_div_factor: float = 1e4, base_momentum: float = 0.85,
             max_momentum: float = 0.95) -> Tuple[float, float]:
    """(lr, beta1) in force at optimiser step `step` (0-based) of torch's OneCycleLR with default
    arguments: cosine annealing, two phases, cycle_momentum=True -- which overwrites Adam's beta1
    and cycles it 0.95 -> 0.85 -> 0.95, so the configured adam_beta1 is ignored."""
    if step >= total_steps:
        raise ValueError(f"Tried to step {step + 1} times. The specified number of total steps is {total_steps}")

    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    initial_lr = max_lr / div_factor
    min_lr = initial_lr / final_div_factor
    end1 = float(pct_start * total_steps) - 1.0
    end2 = float(total_steps) - 1.0
    if step <= end1:
        pct = step / end1 if end1 != 0 else 0.0
        return cos(initial_lr, max_lr, pct), cos(max_momentum, base_momentum, pct)
    pct = (step - end1) / (end2 - end1)
    return cos(max_lr, min_lr, pct), cos(base_momentum, max_momentum, pct)


class FusedAdamOneCycle:
    """optimiser in {"adam", "adamw"} (OPTIMISER_REGISTRY, wrapper.py:29)."""

    def __init__(self, engine, optimiser: str = "adam", lr: float = 1e-3, weight_decay: float = 0.0,
                 adam_beta1: float = 0.9, adam_beta2: float = 0.999, eps: float = 1e-8,
                 num_steps: int = 1000, clip_grad: float = 1.0, world_size: int = 1):
        if optimiser not in ("adam", "adamw"):
            raise KeyError(optimiser)
        self.engine = engine
        self.decoupled = optimiser == "adamw"
        self.lr, self.wd = float(lr), float(weight_decay)
        self.beta1_cfg, self.beta2, self.eps = float(adam_beta1), float(adam_beta2), float(eps)
        self.num_steps, self.clip = int(num_steps), float(clip_grad or 0.0)
        self.world_size = int(world_size)
        self.step_count = 0
        self.growth, self.backoff, self.growth_interval = 2.0, 0.5, 2000      # torch.amp.GradScaler defaults (fp16 mode only)
        self.beta1_pow = 1.0  # product of the (cycled) beta1 is NOT what torch uses: see step()
        dev = engine.dev
        self.hyper = torch.zeros(10, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sumsq_ws = torch.empty(2048, dtype=torch.float32, device=dev)    # afm_sumsq's partial sums (AFM_SUMSQ_PARTIALS)
        # ring of pinned staging rows: the host may run several steps ahead of the stream
        self._host = torch.zeros(32, 10, dtype=torch.float32)
        if dev.type == "cuda":
            self._host = self._host.pin_memory()
        self._row_events = [None] * self._host.shape[0]   # H2D copy of each staging row: fence before its reuse
        self.last_lr, self.last_beta1 = 0.0, 0.0

    def step(self, grads_are_summed_over_ranks: bool = False) -> None:
        """One optimiser step on the accumulated gradients; zeroes them."""
        ps = self.engine.ps
        lr, beta1 = onecycle(self.step_count, self.num_steps, self.lr)
        self.step_count += 1
        t = self.step_count
        # torch: bias_correction1 = 1 - beta1 ** step with the CURRENT (cycled) beta1
        bc1 = 1.0 - beta1 ** t
        bc2 = 1.0 - self.beta2 ** t
        gmult = 1.0 / self.world_size if grads_are_summed_over_ranks else 1.0
        vals = [lr, beta1, self.beta2, self.eps, self.wd, bc1, bc2, self.clip, gmult, 1.0 if self.decoupled else 0.0]
        slot = self.step_count % self._host.shape[0]
        row = self._host[slot]
        if self._row_events[slot] is not None:
            self._row_events[slot].synchronize()   # only ever waits when the host is a whole ring ahead of the stream
        row.copy_(torch.tensor(vals, dtype=torch.float32))
        self.hyper.copy_(row, non_blocking=True)
        if self.hyper.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._row_events[slot] = ev
        self.sumsq.zero_()
        ops.sumsq(ps.grad, self.sumsq, self.sumsq_ws)
        scaler = getattr(self.engine, "scaler", None)
        ops.adam_step(ps.flat, ps.grad, ps.exp_avg, ps.exp_avg_sq, self.hyper, self.sumsq, ps.bf16, zero_grad=True, scaler=scaler)
        if scaler is not None:   # fp16: GradScaler.update() on the device (a skipped step halves S, 2000 good ones double it)
            ops.scaler_update(scaler, self.sumsq, self.growth, self.backoff, self.growth_interval)
        self.engine.refresh_transposes()
        self.last_lr, self.last_beta1 = lr, beta1

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm measured by the last step (device tensor; before clipping).  fp16 mode: the buffer held S x the
        gradients when it was measured; the scaler may have moved since, so this is exact only between scale changes."""
        g = 1.0 / self.world_size if self.world_size > 1 else 1.0
        scaler = getattr(self.engine, "scaler", None)
        if scaler is not None:
            return torch.sqrt(self.sumsq[0]) * g / scaler[0]
        return torch.sqrt(self.sumsq[0]) * g

    def state_dict(self):
        ps = self.engine.ps
        sd = {"step": self.step_count, "exp_avg": ps.exp_avg.clone(), "exp_avg_sq": ps.exp_avg_sq.clone()}
        if getattr(self.engine, "scaler", None) is not None:
            sd["loss_scaler"] = self.engine.scaler.clone()
        return sd

    def load_state_dict(self, sd):
        ps = self.engine.ps
        self.step_count = int(sd["step"])
        ps.exp_avg.copy_(sd["exp_avg"]); ps.exp_avg_sq.copy_(sd["exp_avg_sq"])
        if sd.get("loss_scaler") is not None and getattr(self.engine, "scaler", None) is not None:
            self.engine.scaler.copy_(sd["loss_scaler"].to(self.engine.scaler.device))
