"""Load tests/golden/*.npz (written by oracle/make_goldens.py) into nested dicts of torch tensors."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    tree = {}
    for k in z.files:
        if k == "meta":
            tree["meta"] = json.loads(bytes(z[k]).decode())
            continue
        node = tree
        parts = k.split("/")
        # state-dict style groups keep the dotted name as ONE key
        if parts[0] in ("sd", "grad0", "step1", "step2") or (len(parts) > 1 and parts[1] == "sd"):
            idx = 1 if parts[0] in ("sd", "grad0", "step1", "step2") else 2
            for p in parts[:idx]:
                node = node.setdefault(p, {})
            node["/".join(parts[idx:])] = torch.from_numpy(z[k])
            continue
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = torch.from_numpy(z[k])
    return tree


def batch_of(tree, i):
    b = tree[f"b{i}"]
    return {
        "encoder_input": dict(b["encoder_input"]),
        "encoder_pad_mask": b["encoder_pad_mask"],
        "decoder_input": dict(b["decoder_input"]),
        "decoder_pad_mask": b["decoder_pad_mask"],
        "target": b["target"],
        **({"encoder_alignment_input": b["encoder_alignment_input"]} if "encoder_alignment_input" in b else {}),
    }


def model_cfg(meta):
    """CustomConfig kwargs -> the plain dict the oracle / engine take (defaults of
    custom_modeling.py:43-65 filled in)."""
    cfg = dict(d_model=512, max_position_embeddings=1024, encoder_layers=6, decoder_layers=6,
               encoder_attention_heads=8, decoder_attention_heads=8, encoder_ffn_dim=2048,
               decoder_ffn_dim=2048, dropout=0.1, gated_linear=False,
               positional_encoding_type="sin_cos", multimodal_norm=True)
    cfg.update(meta["cfg"])
    return cfg
