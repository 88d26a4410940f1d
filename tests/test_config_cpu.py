"""CPU: the Hydra-style composer resolves this repo's configs/ tree and, when present (build
container only), the REFERENCE's own configs/ tree with the reference test's override list
(reference tests/test_run.py:8-18) to the same model / data / trainer settings."""
import os

import pytest

from multimodalanalytical_amd.config import compose, wrapper_kwargs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OVERRIDES = ["working_dir=runs", "job_name=train", "data=ir/patches", "data_path=tests/test_data/ir_dataset",
             "data.IR.preprocessor_arguments.patch_size=125", "data.Formula.column=molecular_formula",
             "model=custom_model", "molecules=True", "trainer.epochs=1"]


def _check(cfg):
    assert cfg["trainer"]["epochs"] == 1 and cfg["trainer"]["acc_batches"] == 4 and cfg["trainer"]["clip_grad"] == 1.0
    assert cfg["trainer"]["log_dir"] == "runs" and cfg["trainer"]["task"] == "train"      # ${...} interpolation
    assert cfg["data"]["IR"]["preprocessor_arguments"]["patch_size"] == 125
    assert cfg["data"]["IR"]["type"] == "1D_patches" and cfg["data"]["Smiles"]["target"] is True
    m = wrapper_kwargs(cfg)
    assert m["model_type"] == "CustomModel" and m["d_model"] == 512 and m["encoder_layers"] == 6
    assert m["encoder_ffn_dim"] == 2048 and m["batch_size"] == 128 and m["optimiser"] == "adamw"
    assert m["positional_encoding_type"] == "sin_cos" and m["gated_linear"] is False
    assert cfg["molecules"] is True and cfg["mixture"] is None


def test_compose_own_tree():
    cfg = compose(os.path.join(ROOT, "configs"), "config_train", OVERRIDES)
    _check(cfg)
    cfg = compose(os.path.join(ROOT, "configs"), "config_train",
                  ["data=multimodal/multimodal", "model=custom_model_base", "modality_dropout=[IR,Multiplets,Carbon]",
                   "model.lr=1.e-3", "model.gated_linear=True"])
    assert list(cfg["data"]) == sorted(cfg["data"]) or set(cfg["data"]) == {"Formula", "Multiplets", "Carbon", "IR", "Smiles"}
    assert cfg["modality_dropout"] == ["IR", "Multiplets", "Carbon"]
    assert cfg["model"]["d_model"] == 768 and cfg["model"]["lr"] == 1e-3 and cfg["model"]["gated_linear"] is True


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference tree only exists in the build container")
def test_compose_reference_tree_unchanged():
    cfg = compose("/root/reference/configs", "config_train", OVERRIDES)
    _check(cfg)
    ours = compose(os.path.join(ROOT, "configs"), "config_train", OVERRIDES)
    assert ours["model"] == cfg["model"]
    assert ours["data"] == cfg["data"]
    assert ours["trainer"] == cfg["trainer"]


def test_cli_plan_from_the_reference_tests_override_list():
    """`python -m multimodalanalytical_amd.cli.training <the override list of reference tests/test_run.py:8-18>`:
    everything up to the GPU (compose -> data_config with the tokenised data's vocab sizes -> HFWrapper keywords ->
    optimiser-step count -> paths) resolves and is what the reference's CLI would hand to HFWrapper / build_trainer."""
    from multimodalanalytical_amd.cli.training import build_plan, split_cli
    from multimodalanalytical_amd.synth import synth_shards
    own, ov = split_cli(OVERRIDES + ["precision=bf16", "strict=1"])
    assert own == {"precision": "bf16", "strict": "1"} and ov == OVERRIDES
    cfg = compose(os.path.join(ROOT, "configs"), "config_train", ov)
    shards = synth_shards(cfg["data"], 1000, 16, 16)
    plan = build_plan(cfg, shards["train"]["meta"], 1000)
    assert plan["target_modality"] == "Smiles" and plan["run_dir"] == os.path.join("runs", "train")
    assert plan["data_config"]["Smiles"]["vocab_size"] == 64 and plan["data_config"]["IR"]["preprocessor_arguments"]["patch_size"] == 125
    assert plan["train_steps"] == 2          # ceil(ceil(1000/128)/4) * 1 epoch (utils.py:156-172)
    assert build_plan(cfg, shards["train"]["meta"], 1000, world_size=8)["train_steps"] == 1
    assert build_plan(cfg, shards["train"]["meta"], 1000, world_size=8, legacy_step_count=True)["train_steps"] == 2
    assert plan["monitor"] == "val_molecular_accuracy" and plan["monitor_mode"] == "max" and plan["n_beams"] == 10
    assert plan["model_config"]["model_type"] == "CustomModel" and plan["acc_batches"] == 4 and plan["clip_grad"] == 1.0
    assert shards["train"]["data"]["IR"]["spectra"].shape == (1000, 1800)
    assert shards["train"]["data"]["Smiles"]["input_ids"].shape == (1000, 65)


C5 = ["data=ir/patches_mixture_text_align", "mixture=ir/binary", "model=custom_model_align"]


def test_compose_mixture_run_groups():
    """The groups of the reference's mixture runs (paper_replication/mixture scripts: data=ir/patches_mixture_text_align,
    mixture=ir/binary, model=custom_model_align) exist in this repo's tree with the reference's keys."""
    cfg = compose(os.path.join(ROOT, "configs"), "config_train", C5 + ["mixture.balanced.train_max_n_samples=4096"])
    assert cfg["mixture"]["balanced"]["n_compounds"] == 2 and cfg["mixture"]["balanced"]["train_max_n_samples"] == 4096
    assert cfg["data"]["IR_target"]["alignment"] is True and cfg["data"]["IR_target"]["target"] is True
    assert cfg["model"]["align_config"]["loss_lambda"] == 50 and cfg["model"]["align_config"]["align_network"] == "convolutional"
    from multimodalanalytical_amd.cli.training import build_plan
    from multimodalanalytical_amd.synth import synth_shards
    shards = synth_shards(cfg["data"], 64, 8, 8)
    assert "IR_target" not in shards["train"]["data"]                       # made by the mixture generator, not stored
    plan = build_plan(cfg, shards["train"]["meta"], 4096)
    assert plan["target_modality"] == "Smiles" and plan["model_config"]["align_config"]["output_dimension"] == 1800


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference tree only exists in the build container")
def test_mixture_run_groups_equal_the_reference_tree():
    ours = compose(os.path.join(ROOT, "configs"), "config_train", C5)
    ref = compose("/root/reference/configs", "config_train", C5)
    assert ours["data"] == ref["data"] and ours["mixture"] == ref["mixture"] and ours["model"] == ref["model"]
