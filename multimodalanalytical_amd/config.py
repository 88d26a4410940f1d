"""Minimal Hydra-compatible config composer (Hydra / OmegaConf are not installed in this image).

Keeps the reference's config SURFACE (SURVEY 8b "Config compatibility"): a root yaml with a
`defaults:` list selecting one file per config group (`data: ir/patches`, `model: [custom_model]`,
`group: null`), `# @package _global_`, `${key}` / `${a.b}` interpolation, and the command-line
override grammar its scripts use (reference scripts/*.sh, tests/test_run.py:8-18):
`group=option`, `a.b.c=value` (YAML-typed), `key=[x,y]`.  The same function composes the
reference's own `configs/` tree unchanged (pointed at with `config_dir`) and this repo's `configs/`.
"""
from __future__ import annotations

import copy
import os
import re
from typing import Any, Dict, List, Optional

import yaml

_INTERP = re.compile(r"\$\{([^}]+)\}")


_FLOAT = re.compile(r"[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+")


def _typed(v: Any) -> Any:
    """OmegaConf reads `1e-4` as a float; PyYAML (YAML 1.1) leaves it a string."""
    if isinstance(v, str) and _FLOAT.fullmatch(v.strip()):
        return float(v)
    if isinstance(v, dict):
        return {k: _typed(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_typed(x) for x in v]
    return v


def _load(path: str) -> Dict[str, Any]:
    with open(path) as fh:
        return _typed(yaml.safe_load(fh) or {})


def _merge(dst: Dict[str, Any], src: Dict[str, Any]) -> Dict[str, Any]:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set(cfg: Dict[str, Any], dotted: str, value: Any) -> None:
    node = cfg
    parts = dotted.split(".")
    for p in parts[:-1]:
        if not isinstance(node.get(p), dict):
            node[p] = {}
        node = node[p]
    node[parts[-1]] = value


def _get(cfg: Dict[str, Any], dotted: str) -> Any:
    node: Any = cfg
    for p in dotted.split("."):
        node = node[p]
    return node


def _resolve(cfg: Dict[str, Any]) -> None:
    def walk(node):
        it = node.items() if isinstance(node, dict) else enumerate(node)
        for k, v in list(it):
            if isinstance(v, (dict, list)):
                walk(v)
            elif isinstance(v, str) and "${" in v:
                def sub(m):
                    key = m.group(1)
                    if key.startswith("now:"):
                        return m.group(0)            # hydra run-dir timestamps: left to the launcher
                    val = _get(cfg, key)
                    return "" if val is None else str(val)
                whole = _INTERP.fullmatch(v)
                if whole and not whole.group(1).startswith("now:"):
                    node[k] = _get(cfg, whole.group(1))
                else:
                    node[k] = _INTERP.sub(sub, v)
    walk(cfg)


def compose(config_dir: str, config_name: str = "config_train", overrides: Optional[List[str]] = None) -> Dict[str, Any]:
    overrides = list(overrides or [])
    root = _load(os.path.join(config_dir, config_name + ".yaml"))
    defaults = root.pop("defaults", [])
    groups: Dict[str, Any] = {}
    order: List[str] = []
    for d in defaults:
        if d == "_self_":
            order.append("_self_")
        elif isinstance(d, dict):
            (g, opt), = d.items()
            groups[g] = opt
            order.append(g)
    plain = []
    for ov in overrides:                               # group selections first
        key, _, val = ov.partition("=")
        key = key.lstrip("+")
        if key in groups and "." not in key:
            groups[key] = yaml.safe_load(val)
        else:
            plain.append((key, val))
    cfg: Dict[str, Any] = {}
    for item in order:
        if item == "_self_":
            _merge(cfg, root)
            continue
        opt = groups[item]
        if opt is None:
            cfg.setdefault(item, None)
            continue
        for o in (opt if isinstance(opt, list) else [opt]):
            path = os.path.join(config_dir, item, str(o) + ".yaml")
            if not os.path.exists(path):
                raise FileNotFoundError(f"config group option {item}={o}: {path} not found")
            sub = _load(path)
            if item == "hydra":
                cfg.setdefault("hydra", {})
                _merge(cfg["hydra"], sub)
            elif isinstance(cfg.get(item), dict):
                _merge(cfg[item], sub)
            else:
                cfg[item] = copy.deepcopy(sub)
    for key, val in plain:
        _set(cfg, key, _typed(yaml.safe_load(val)) if val != "" else None)
    _resolve(cfg)
    return cfg


def wrapper_kwargs(cfg: Dict[str, Any]) -> Dict[str, Any]:
    """The `**model_config` splat of the reference (cli/training.py:139-145): the model yaml goes
    verbatim into HFWrapper(...)."""
    return dict(cfg["model"])
