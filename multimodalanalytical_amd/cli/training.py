"""Config-driven training entry: the surface of the reference's `cli/training.py` (reference cli/training.py:44-254)
on the HIP path.

    python -m multimodalanalytical_amd.cli.training working_dir=runs job_name=train data=ir/patches \
        data_path=<shard dir> model=custom_model trainer.epochs=1            # same key=value grammar as the reference

    compose(configs/config_train.yaml + overrides)  ->  data_config / model_config / trainer settings
    -> PatchPreprocessor statistics from the train shard, DeviceCollator            (reference: load_preprocessors,
    -> calculate_training_steps -> HFWrapper(data_config, target_tokenizer, num_steps,     MultiModalDataModule)
                                             modality_dropout, **model_config)
    -> TrainLoop (accumulate / clip / AdamW + OneCycle, RCCL data parallel when launched with torch.distributed.run)
    -> validation every epoch, last.ckpt + top-5 checkpoints by `trainer.checkpoint_monitor`, best.ckpt
    -> reload best, predict the test shard with beam search, metrics_beam_{n}_{rank}.json (Top-k exact-sequence accuracy)

What is NOT here (out of scope): parquet ETL, tokenizer training, RDKit canonicalisation.  `data_path`
therefore points at PRE-TOKENISED shards, `{train,val,test}.pt`, each `{"meta": {modality: {"vocab_size", "pad_token_id"}},
"data": {modality: {"input_ids", "attention_mask"} | {"spectra"[, "present"]}}}` (text-like modalities as padded id
matrices, patch modalities as raw spectra; `multimodalanalytical_amd.synth.write_shards` writes synthetic ones), or
`data_path=synthetic:<n_train>` to generate them on the fly from the composed data config.
Like the reference the run is wrapped in one try/except that logs and returns 0 unless `strict=1` is passed.
"""
from __future__ import annotations

import json
import math
import os
import shutil
import sys
import traceback
from typing import Any, Dict, List, Optional

import torch

from ..config import compose, wrapper_kwargs
from ..params import PATCH_TYPES, TEXT_TYPES
from ..trainer import barrier, sync_flag, calculate_training_steps

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(HERE)), "configs")


def split_cli(argv: List[str]):
    """Pseudo-overrides consumed here (not part of the Hydra tree): config_dir=, config_name=, strict=, precision=, device=."""
    own, rest = {}, []
    for a in argv:
        k, _, v = a.partition("=")
        if k in ("config_dir", "config_name", "strict", "precision", "device", "max_steps"):
            own[k] = v
        else:
            rest.append(a)
    return own, rest


def build_plan(cfg: Dict[str, Any], shard_meta: Dict[str, Dict[str, Any]], n_train: int, world_size: int = 1,
               legacy_step_count: bool = False) -> Dict[str, Any]:
    """Everything the run needs that does not touch the GPU: the model's data_config (config group `data` + the vocab
    sizes / pad ids only the tokenised data knows), HFWrapper keywords, optimiser-step count, output paths."""
    data_config: Dict[str, Any] = {}
    target = None
    for m, mc in cfg["data"].items():
        mc = dict(mc)
        if mc["type"] in TEXT_TYPES:
            if m not in shard_meta:
                raise KeyError(f"modality {m!r} of the data config is missing from the shards")
            mc["vocab_size"] = int(shard_meta[m]["vocab_size"])
            mc["pad_token_id"] = int(shard_meta[m].get("pad_token_id", 0))
        elif mc["type"] not in PATCH_TYPES:
            raise NotImplementedError(f"modality type {mc['type']!r}")
        if mc.get("target") and not mc.get("alignment"):      # (an alignment modality is a target too, datamodules.py:45-56)
            target = m
        data_config[m] = mc
    if target is None:
        raise ValueError("no target modality in the data config")
    model_cfg = wrapper_kwargs(cfg)
    tr = cfg["trainer"]
    steps = calculate_training_steps(n_train, model_cfg["batch_size"], tr["acc_batches"], tr["epochs"],
                                     1 if legacy_step_count else world_size)   # reference utils.py:164-167 hard-codes 1 GPU
    run_dir = os.path.join(str(cfg["working_dir"]), str(cfg["job_name"]))
    return {"data_config": data_config, "target_modality": target, "model_config": model_cfg, "train_steps": steps,
            "modality_dropout": cfg.get("modality_dropout"), "run_dir": run_dir,
            "n_beams": model_cfg.get("n_beams", 10), "monitor": tr["checkpoint_monitor"],
            "monitor_mode": "min" if "loss" in tr["checkpoint_monitor"] else "max",          # trainer/trainer.py:34
            "acc_batches": tr["acc_batches"], "clip_grad": tr["clip_grad"], "epochs": tr["epochs"],
            "limit_val_batches": tr.get("limit_val_batches", 1.0), "early_stopping_patience": tr.get("early_stopping_patience")}


# ------------------------------------------------------------------------------------------------ shards
def load_shards(data_path: str, cfg: Dict[str, Any], device: str):
    if str(data_path).startswith("synthetic:"):
        from ..synth import synth_shards
        n = int(str(data_path).split(":", 1)[1])
        return synth_shards(cfg["data"], n, max(8, n // 8), max(8, n // 8), seed=3247)
    out = {}
    for split in ("train", "val", "test"):
        p = os.path.join(data_path, split + ".pt")
        if os.path.exists(p):
            out[split] = torch.load(p, map_location="cpu", weights_only=False)
    if "train" not in out:
        raise FileNotFoundError(f"{data_path}/train.pt not found (pre-tokenised shards, see the module docstring)")
    return out


def shard_len(shard) -> int:
    first = next(iter(shard["data"].values()))
    return int((first["input_ids"] if "input_ids" in first else first["spectra"]).shape[0])


class ShardLoader:
    """Rank-strided, seeded-permutation batches of one shard, collated on the device (DistributedSampler semantics:
    every rank sees a disjoint 1/world of each epoch; the tail that does not fill a batch on every rank is dropped for
    training so all ranks take the same number of optimiser steps)."""

    def __init__(self, shard, collator, batch_size: int, device: str, rank: int = 0, world: int = 1, shuffle: bool = True,
                 drop_last: bool = True):
        self.shard, self.collator, self.bs, self.dev = shard, collator, int(batch_size), device
        self.rank, self.world, self.shuffle, self.drop_last = rank, world, shuffle, drop_last
        self.n = shard_len(shard)
        self.dev_data = {m: {k: v.to(device) for k, v in d.items()} for m, d in shard["data"].items()}   # resident in HBM
        # AFM_SORT_BATCH=1 / 2 (training shuffles only; A / B probes, OFF by default): the samples of a micro-batch in order of decreasing
        # encoder length (1), or that order dealt in a snake over the batch's eight contiguous eighths (2).  Which samples share a batch
        # does not change and neither does the loss.  Measured: 1 loses 4 ... 6 % on the padded workloads (the
        # live rows pile up in the first XCDs' row ranges of the hinted backward GEMMs), 2 equals the random order.
        self.sort_len = None
        self.sort_mode = os.environ.get("AFM_SORT_BATCH", "0")
        if shuffle and self.sort_mode in ("1", "2"):
            inputs = set(getattr(collator, "input_modalities", []))
            masks = [d["attention_mask"].sum(1) for m, d in self.dev_data.items() if m in inputs and "attention_mask" in d]
            if masks:
                self.sort_len = torch.stack(masks, 0).sum(0)

    def __len__(self):
        per_rank = self.n // self.world
        return per_rank // self.bs if self.drop_last else math.ceil(per_rank / self.bs)

    def epoch(self, epoch: int):
        g = torch.Generator().manual_seed(3247 + epoch)
        order = torch.randperm(self.n, generator=g) if self.shuffle else torch.arange(self.n)
        order = order[:(self.n // self.world) * self.world][self.rank::self.world].to(self.dev)
        for i in range(len(self)):
            idx = order[i * self.bs:(i + 1) * self.bs]
            if self.sort_len is not None:
                idx = idx[torch.argsort(self.sort_len[idx], descending=True, stable=True)]
                if self.sort_mode == "2" and idx.numel() % 8 == 0:
                    # snake deal: the sorted samples go round the 8 contiguous eighths of the batch (= the row ranges the persistent
                    # kernels give an XCD each), forwards and backwards in turn, so every eighth holds the same share of live rows
                    n8 = idx.numel() // 8
                    j = torch.arange(8, device=idx.device)[:, None]; r = torch.arange(n8, device=idx.device)[None, :]
                    idx = idx[(8 * r + torch.where(r % 2 == 0, j, 7 - j)).reshape(-1)]
            yield self.collator({m: {k: v.index_select(0, idx) for k, v in d.items()} for m, d in self.dev_data.items()})


class MixtureLoader:
    """The reference's mixture datasets (data/datasets.py:395-412: with a `mixture` config group every split becomes an iterable of
    mixtures, `multi_config_mix` over the named configurations) on the device: the shard holds the PURE compounds (ids of the text
    modalities, IR spectra) resident in HBM, every configuration is a `MixtureGenerator` (index stream on the host with the
    reference's numpy RNG, the weighted average in `afm_mix_spectra`, rank-strided under data parallelism), the configurations
    alternate record by record, and the records are cut into batches and collated:

        IR          the mixed spectrum                      (input modality)
        IR_target   the pure spectrum of the record's compound -> `encoder_alignment_input` (the alignment head's target)
        text ids    gathered from the compound's row

    Like the reference's generator it restarts with the same seed every epoch, and its nominal length is the configured
    sum of `<split>_max_n_samples` (what `calculate_training_steps` is given), not the number of records that survive the
    duplicate / permutation checks."""

    SPLIT_KEY = {"train": "train", "val": "validation", "test": "test"}

    def __init__(self, shard, mixture_cfg, split, collator, batch_size: int, device: str, rank: int = 0, world: int = 1,
                 spectrum_modality: str = "IR", seed: int = 3247):
        from ..preprocess import MixtureGenerator
        self.collator, self.bs, self.dev, self.rank, self.world = collator, int(batch_size), device, rank, world
        self.key = self.SPLIT_KEY[split]
        self.spec = spectrum_modality
        self.dev_data = {m: {k: v.to(device) for k, v in d.items()} for m, d in shard["data"].items()}
        table = self.dev_data[spectrum_modality]["spectra"]
        self.configs = {name: dict(c) for name, c in mixture_cfg.items() if c.get(f"{self.key}_max_n_samples", 0) > 0 or c.get("mixed")}
        self.gens = [MixtureGenerator(table, c, self.key, seed, rank, world) for c in self.configs.values()]
        self.nominal = sum(int(c.get(f"{self.key}_max_n_samples", 0)) for c in mixture_cfg.values())
        self.align = collator.alignment_modality[0] if collator.alignment_modality else None

    def __len__(self):
        """Batches of one pass at the NOMINAL length: training drops a trailing partial batch, validation / test keep it (as epoch()
        below, and as ShardLoader with drop_last=False)."""
        per_rank = self.nominal // self.world
        return per_rank // self.bs if self.key == "train" else math.ceil(per_rank / self.bs)

    def records(self):
        from ..preprocess import interleave_rounds
        yield from interleave_rounds(self.gens)

    def epoch(self, epoch: int):     # noqa: ARG002 (the reference's generator reseeds: every epoch is the same stream)
        """Batches of one pass over the stream.  The reference's DataLoaders all run with drop_last=False (data/datamodules.py:433,467,
        502).  Validation / test do the same here: the trailing records are a final short batch, every record is evaluated.  TRAINING
        drops a trailing partial batch -- a deliberate difference, shared with ShardLoader: every rank takes the same number of
        optimiser steps and every micro-batch has the shape the kernels were planned for; at most batch_size - 1 records of an epoch's
        stream are not seen.  A split that yields nothing at all is an error here, not a NaN monitor score later."""
        pend, n = None, 0
        for r in self.records():
            r = {k: v for k, v in r.items() if k in ("IR", "compound", "IR_target")}
            pend = r if pend is None else {k: torch.cat([pend[k], r[k]]) for k in r}
            while pend["compound"].shape[0] >= self.bs:
                cut = {k: v[:self.bs] for k, v in pend.items()}
                pend = {k: v[self.bs:] for k, v in pend.items()}
                n += 1
                yield self._collate(cut)
        if self.key != "train" and pend is not None and pend["compound"].shape[0] > 0:
            n += 1
            yield self._collate(pend)
        if n == 0:
            raise RuntimeError(f"mixture split '{self.key}' produced no batch (nominal {self.nominal} samples over {self.world} rank(s), "
                               f"batch size {self.bs}): the table is too small for this mixture configuration")

    def _collate(self, rec):
        comp = rec["compound"]
        inputs = {}
        for m, d in self.dev_data.items():
            if m == self.spec:
                inputs[m] = {"spectra": rec["IR"]}
            else:
                inputs[m] = {k: v.index_select(0, comp) for k, v in d.items()}
        if self.align is not None:
            inputs[self.align] = {"spectra": rec["IR_target"]}
        return self.collator(inputs)


def val_batches(loader, limit):
    """(index, batch) of one validation pass under Lightning's `limit_val_batches` (an int = that many batches, a float <= 1 = that
    share of them, at least one; trainer/trainer.py:66).  The count comes from len(loader), which for validation INCLUDES a trailing
    short batch (drop_last=False in the reference, data/datamodules.py:467); a split without a single batch raises here, before the
    loop, instead of leaving on_validation_epoch_end without outputs."""
    total = len(loader)
    if total <= 0:
        raise RuntimeError("the validation split yields no batch (fewer records than ranks, or an empty mixture configuration)")
    nval = min(total, int(limit) if isinstance(limit, int) or float(limit) > 1.0 else max(1, int(total * float(limit))))
    for i, batch in enumerate(loader.epoch(0)):
        if i >= nval:
            break
        yield i, batch


def build_preprocessors(train_shard, data_config, device, mixture_sample=None):
    """`load_preprocessors` for the patch modalities (reference data/data_utils.py -> PatchPreprocessor.initialise):
    statistics over the non-zero entries of (a sample of) the training spectra."""
    from ..preprocess import PatchPreprocessor
    pre = {}
    for m, mc in data_config.items():
        if mc["type"] == "1D_patches":
            a = mc.get("preprocessor_arguments") or {}
            pp = PatchPreprocessor(patch_size=int(a["patch_size"]), masking=bool(a.get("masking", False)),
                                   interpolation=bool(a.get("interpolation", False)), overlap=int(a.get("overlap", 1)),
                                   derivative=bool(a.get("derivative", False)), device=device)
            # data_utils.py:49-59: statistics from (up to) the first 10 000 records -- of the mixture stream when there is one
            # (`data_set.take(num_samples)`), else a sample of the training shard
            if mixture_sample is not None and m in mixture_sample:
                sample = mixture_sample[m]
            elif m in train_shard["data"]:
                sample = train_shard["data"][m]["spectra"][:10000]
            else:
                sample = None       # (an alignment modality the shards do not carry: only its `interpolation` flag is consulted)
            if sample is not None:
                pp.initialise({m: sample.cpu().numpy()}, m)
            pre[m] = pp
    return pre


def setup_data(cfg: Dict[str, Any], device: str, world: int = 1):
    """Shards, run plan, collator and the mixture group of a composed config (the reference's build_dataset_multimodal +
    load_preprocessors + MultiModalDataModule, cli/training.py:80-131)."""
    from ..preprocess import DeviceCollator
    shards = load_shards(cfg["data_path"], cfg, device)
    meta = shards["train"]["meta"]
    mixture = cfg.get("mixture")            # cli/training.py:86-94 hands it to the dataset builder (datasets.py:395-412)
    n_train = shard_len(shards["train"])
    if isinstance(mixture, dict):
        n_train = sum(int(c.get("train_max_n_samples", 0)) for c in mixture.values())      # the iterable's nominal length
    plan = build_plan(cfg, meta, n_train, world, bool(cfg["trainer"].get("legacy_step_count", False)))
    dc, tm = plan["data_config"], plan["target_modality"]
    mix_sample = None
    if isinstance(mixture, dict):           # preprocessor statistics from the head of the mixture stream (data_utils.py:49-54)
        probe = MixtureLoader(shards["train"], mixture, "train", DeviceCollator(dc, {}, tm), 1, device, 0, 1)
        got, rows = [], 0
        for r in probe.records():
            got.append(r["IR"]); rows += int(r["IR"].shape[0])
            if rows >= 10000:
                break
        if got:
            mix_sample = {probe.spec: torch.cat(got)[:10000]}
        del probe
    pre = build_preprocessors(shards["train"], dc, device, mix_sample)
    return shards, plan, DeviceCollator(dc, pre, tm), (mixture if isinstance(mixture, dict) else None)


# ------------------------------------------------------------------------------------------------ run
def topk_sequence_accuracy(model, predictions: torch.Tensor, targets: torch.Tensor, n_beams: int) -> Dict[str, float]:
    """`calc_sampling_metrics(..., molecules=False)` on token ids: Top-k = a target equals one of the first k beams."""
    B = targets.shape[0]
    hit_at = torch.full((B,), n_beams, dtype=torch.long)
    for k in range(n_beams):
        rows = predictions.view(B, n_beams, -1)[:, k]
        same = torch.stack([model.sequence_accuracy(rows[i:i + 1], targets[i:i + 1])[0] for i in range(B)]).cpu() > 0.5
        hit_at = torch.where(same & (hit_at == n_beams), torch.full_like(hit_at, k), hit_at)
    return {f"Top-{k + 1}": float((hit_at <= k).float().mean()) for k in range(n_beams)}


def run(cfg: Dict[str, Any], own: Dict[str, str]) -> Dict[str, Any]:
    import torch.distributed as dist
    from ..modeling.wrapper import HFWrapper, SimpleTokenizerInfo
    from ..preprocess import DeviceCollator
    from ..trainer import TrainLoop, load_checkpoint, save_checkpoint
    from ..x2 import X2
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = own.get("device", f"cuda:{local}")
    torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=torch.device(device))
    shards, plan, collator, mixture = setup_data(cfg, device, world)
    os.makedirs(plan["run_dir"], exist_ok=True)
    dc, tm = plan["data_config"], plan["target_modality"]
    tok = SimpleTokenizerInfo(dc[tm]["vocab_size"], pad_token_id=dc[tm]["pad_token_id"])
    # precision=: fp16 (= Lightning's "16-mixed", what the reference trains with on a GPU, trainer/trainer.py:69: single-pass fp16
    # MFMA forward and backward, dynamic loss scaling; logits within the 1e-3 bar) | bf16x3-mixed (split-pair forward = fp32-grade
    # outputs, single-pass bf16 backward) | bf16x3 (split pairs in both directions) | bf16 (= "bf16-mixed") | fp32 (= "32-true")
    precision = {"16-mixed": "fp16", "bf16-mixed": "bf16", "32-true": "fp32", "32": "fp32"}.get(str(own.get("precision", "fp16")),
                                                                                                 str(own.get("precision", "fp16")))
    cd = {"fp16": torch.float16, "bf16x3-mixed": X2.dtype, "bf16x3": X2.dtype, "bf16": torch.bfloat16, "fp32": torch.float32}[precision]
    bd = torch.bfloat16 if precision == "bf16x3-mixed" else None
    mk = {k: v for k, v in plan["model_config"].items() if k != "multimodal_norm"}
    bs = int(mk["batch_size"])

    def new_model():
        return HFWrapper(dc, target_tokenizer=tok, num_steps=plan["train_steps"], modality_dropout=plan["modality_dropout"],
                         multimodal_norm=plan["model_config"].get("multimodal_norm", True), clip_grad=plan["clip_grad"],
                         world_size=world, device=device, compute_dtype=cd, backward_dtype=bd, **mk)
    model = new_model()
    loop = TrainLoop(model, acc_batches=plan["acc_batches"], world_size=world)
    if mk.get("model_checkpoint_path"):
        load_checkpoint(mk["model_checkpoint_path"], model, None if cfg.get("finetuning") else loop)
    if isinstance(mixture, dict):
        train = MixtureLoader(shards["train"], mixture, "train", collator, bs, device, rank, world)
        val = MixtureLoader(shards.get("val", shards["train"]), mixture, "val", collator, bs, device, rank, world)
    else:
        train = ShardLoader(shards["train"], collator, bs, device, rank, world, shuffle=True)
        val = ShardLoader(shards.get("val", shards["train"]), collator, bs, device, rank, world, shuffle=False, drop_last=False)
    ckpt_dir = os.path.join(plan["run_dir"], "checkpoints")
    os.makedirs(ckpt_dir, exist_ok=True)
    top: List[tuple] = []        # (score, path), best first
    sign = 1.0 if plan["monitor_mode"] == "max" else -1.0
    max_steps = int(own.get("max_steps", 0))
    history, stale = [], 0
    first_logged = None
    for epoch in range(plan["epochs"]):
        for i, batch in enumerate(train.epoch(epoch)):
            if loop.optim.step_count >= plan["train_steps"] or (max_steps and loop.optim.step_count >= max_steps):
                break
            loop.micro_batch(batch, i)
            if first_logged is None:      # what Lightning would log at step 0: train_loss (+ model_only / alignment loss with the head)
                first_logged = {k: float(v) for k, v in model.logged.items() if k.startswith("train_")}
        loop.flush()      # Lightning steps the optimiser on the last batch of an epoch even when the accumulation window is not full
        for i, batch in val_batches(val, plan["limit_val_batches"]):
            model.validation_step(batch, i)
        avg = {k: float(v) for k, v in model.on_validation_epoch_end().items()}
        history.append({"epoch": epoch, "step": loop.optim.step_count, **avg})
        if rank == 0:
            score = avg.get(plan["monitor"], float("nan"))
            path = os.path.join(ckpt_dir, f"epoch_{epoch}-step_{loop.optim.step_count}.ckpt")
            save_checkpoint(path, model, loop, epoch)
            shutil.copy(path, os.path.join(ckpt_dir, "last.ckpt"))                      # save_last=True
            top.append((sign * score if score == score else -float("inf"), path))
            top.sort(key=lambda t: -t[0])
            for _, old in top[5:]:                                                      # save_top_k=5
                if os.path.exists(old):
                    os.remove(old)
            top[:] = top[:5]
            stale = 0 if top[0][1] == path else stale + 1
        # EarlyStopping stops every rank in the same epoch: rank 0 owns the checkpoint ranking, its decision is broadcast
        if sync_flag(bool(plan["early_stopping_patience"] and stale >= plan["early_stopping_patience"]), 0, device):
            break
    result = {"history": history, "run_dir": plan["run_dir"], "train_steps": plan["train_steps"], "precision": precision,
              "first_train_step": first_logged, "optimizer_steps": loop.optim.step_count}
    if rank == 0:
        best = top[0][1]
        shutil.copy(best, os.path.join(ckpt_dir, "best.ckpt"))
        result["best_model_path"] = best
    if world > 1:
        barrier()
    # reload the best model and evaluate the test shard with beam search (cli/training.py:167-249)
    best_model = new_model()
    load_checkpoint(os.path.join(ckpt_dir, "best.ckpt"), best_model)
    best_model.eval()
    n_beams = int(plan["n_beams"])
    tshard = shards.get("test", shards.get("val", shards["train"]))
    test = (MixtureLoader(tshard, mixture, "test", collator, bs, device, rank, world) if isinstance(mixture, dict) else
            ShardLoader(tshard, collator, bs, device, rank, world, shuffle=False, drop_last=False))
    losses, preds, tgts = [], [], []
    for i, batch in enumerate(test.epoch(0)):
        out = best_model.forward(batch)
        losses.append(float(out.loss))
        seqs = best_model.generate(batch, n_beams=n_beams)
        width = best_model.max_length
        preds.append(torch.nn.functional.pad(seqs, (0, width - seqs.shape[1]), value=tok.pad_token_id))
        tgts.append(batch["target"].T)
        if max_steps and i + 1 >= max_steps:
            break
    metrics = {"avg_loss": sum(losses) / max(1, len(losses))}
    if preds:
        metrics.update(topk_sequence_accuracy(best_model, torch.cat(preds), torch.cat(tgts), n_beams))
    with open(os.path.join(plan["run_dir"], f"metrics_beam_{n_beams}_{rank}.json"), "w") as fh:
        json.dump(metrics, fh)
    result["metrics"] = metrics
    return result


def main(argv: Optional[List[str]] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    own, overrides = split_cli(argv)
    cfg = compose(own.get("config_dir", os.environ.get("AFM_CONFIG_DIR", DEFAULT_CONFIG_DIR)), own.get("config_name", "config_train"),
                  overrides)
    try:
        res = run(cfg, own)
        print(json.dumps({"run_dir": res["run_dir"], "metrics": res["metrics"], "history": res["history"][-1:],
                          "first_train_step": res["first_train_step"], "optimizer_steps": res["optimizer_steps"]}))
    except Exception:       # cli/training.py:253-254: the reference logs the failure and still exits 0
        traceback.print_exc()
        return 1 if own.get("strict", "0") not in ("0", "false", "False") else 0
    return 0


if __name__ == "__main__":
    sys.exit(main())
