"""Golden vectors of the reference's mixture generator (data/datasets.py:49-141: normalize_spectrum, mix_spectra), written to
tests/golden/mixture.npz.  Run in the build container only (it reads /root/reference); the fixture travels, this script's
input does not.

`analytical_fm.data.datasets` cannot be imported here: its module header pulls in omegaconf, pydantic-settings and
pytorch-lightning, none of which is installed (no wheel, no network).  The two functions themselves need numpy, math and a
HF `datasets.Dataset` only, so this script takes their definitions out of the reference's source file with `ast` AT RUN TIME and
executes exactly those (nothing of the file is stored here or in the fixture), with the two module-level names they mention
(`DEFAULT_SETTINGS.default_seed`, configuration.py:10, and the annotation-only `DictConfig`) supplied by hand.  What comes out
is the reference's own arithmetic and the reference's own use of numpy's global RNG on a seeded synthetic table."""
import ast
import math
import os
import sys
import types
from typing import Any, Dict, Generator, List

import numpy as np

REF = "/root/reference/src/analytical_fm/data/datasets.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mixture.npz")


def reference_functions():
    import datasets as hf
    tree = ast.parse(open(REF).read())
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("normalize_spectrum", "mix_spectra")]
    assert [n.name for n in wanted] == ["normalize_spectrum", "mix_spectra"]
    ns = {"np": np, "math": math, "List": List, "Dict": Dict, "Any": Any, "Generator": Generator, "Dataset": hf.Dataset,
          "DictConfig": dict, "DEFAULT_SETTINGS": types.SimpleNamespace(default_seed=3247)}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), REF, "exec"), ns)
    return ns["normalize_spectrum"], ns["mix_spectra"], hf.Dataset


def main():
    normalize_spectrum, mix_spectra, Dataset = reference_functions()
    rng = np.random.default_rng(3247)
    out = {}
    cases = [  # (tag, rows, spectrum length, config)
        ("pair_equal_norm", 12, 1800, dict(n_compounds=2, compounds_ratio=None, parallel_samples=8, train_max_n_samples=32, normalize=True)),
        ("pair_73_raw", 12, 1800, dict(n_compounds=2, compounds_ratio=[0.7, 0.3], parallel_samples=8, train_max_n_samples=24, normalize=False)),
        ("triple_short_norm", 9, 1500, dict(n_compounds=3, compounds_ratio=[0.5, 0.25, 0.25], parallel_samples=6, train_max_n_samples=18, normalize=True)),
        ("pair_zero_weight", 10, 1800, dict(n_compounds=2, compounds_ratio=[1.0, 0.0], parallel_samples=5, train_max_n_samples=10, normalize=True)),
        ("few_samples", 8, 1800, dict(n_compounds=2, compounds_ratio=None, parallel_samples=16, train_max_n_samples=4, normalize=True)),
        ("mixed_passthrough", 6, 1800, dict(n_compounds=2, compounds_ratio=None, parallel_samples=4, train_max_n_samples=8, normalize=True, mixed=True)),
    ]
    names = []
    for tag, n, L, cfg in cases:
        table = rng.standard_normal((n, L)) * 0.3 + 0.4        # some negative points: normalize_spectrum clips after min / max
        table[1] = 0.25                                            # a flat row (max - min == 0 on its own)
        table = table.astype(np.float32).astype(np.float64)      # values a float32 table on the device can hold exactly
        ds = Dataset.from_dict({"Smiles": [f"S{i}" for i in range(n)], "Formula": [f"F{i}" for i in range(n)],
                                "IR": [row.tolist() for row in table]})
        recs = list(mix_spectra(dataset=ds, mix_config=cfg, split="train", seed=3247))
        out[f"{tag}/table"] = table.astype(np.float32)
        out[f"{tag}/cfg_n_compounds"] = np.int64(cfg["n_compounds"])
        out[f"{tag}/cfg_ratio"] = np.asarray(cfg["compounds_ratio"] if cfg["compounds_ratio"] is not None else [], dtype=np.float64)
        out[f"{tag}/cfg_parallel"] = np.int64(cfg["parallel_samples"])
        out[f"{tag}/cfg_max_n"] = np.int64(cfg["train_max_n_samples"])
        out[f"{tag}/cfg_normalize"] = np.bool_(cfg["normalize"])
        out[f"{tag}/cfg_mixed"] = np.bool_(cfg.get("mixed", False))
        out[f"{tag}/ir"] = np.asarray([r["IR"] for r in recs], dtype=np.float64).reshape(len(recs), -1)
        out[f"{tag}/ir_target"] = np.asarray([r["IR_target"] for r in recs], dtype=np.float64).reshape(len(recs), -1)
        out[f"{tag}/smiles"] = np.asarray([int(r["Smiles"][1:]) for r in recs], dtype=np.int64)
        out[f"{tag}/additional"] = np.asarray([r["Additional_smiles"] for r in recs])
        out[f"{tag}/percentage"] = np.asarray([r["Percentage"] for r in recs])
        names.append(tag)
        print(tag, "records", len(recs))
    # normalize_spectrum alone, on lists with negatives / a flat list
    xs = [rng.standard_normal(50).tolist(), [0.5] * 7, (rng.standard_normal(30) - 2.0).tolist()]
    for i, x in enumerate(xs):
        out[f"normalize/{i}/in"] = np.asarray(x, dtype=np.float64)
        out[f"normalize/{i}/out"] = np.asarray(normalize_spectrum(x), dtype=np.float64)
    out["cases"] = np.asarray(names)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    sys.exit(main())
