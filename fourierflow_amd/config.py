"""Config surface of the reference's experiments without Hydra / OmegaConf / Lightning.

The reference composes ``experiments/**/config.yaml`` with Hydra, resolves the custom OmegaConf resolvers
``get_method`` / ``import`` / ``eval`` / ``oc.env`` (fourierflow/__init__.py:19-24) and instantiates
``_target_`` nodes (commands/train.py:38-67).  This module does the same for the part of a config that
drives the hot path -- ``routine`` with its ``conv`` / ``model``, ``optimizer`` and ``scheduler`` nodes -- and
maps ``fourierflow.*`` targets onto their MI355X-native mirrors, so an unmodified torus_li config builds
the HIP-backed routine.  ``builder`` / ``trainer`` / ``callbacks`` / ``wandb`` sections are parsed but not
instantiated (data loading and the Lightning control plane are out of scope).
"""
from __future__ import annotations

import importlib
import os
import re
from typing import Any, Dict, Sequence

import yaml

TARGET_MAP = {
    "fourierflow.modules.FNOFactorized2DBlock": "fourierflow_amd.modules.FNOFactorized2DBlock",
    "fourierflow.modules.FNOFactorizedMesh3D": "fourierflow_amd.modules.FNOFactorizedMesh3D",
    "fourierflow.modules.FNOFactorizedMesh2D": "fourierflow_amd.modules.FNOFactorizedMesh2D",
    "fourierflow.modules.FNOPlus2DBlock": "fourierflow_amd.modules.FNOPlus2DBlock",
    "fourierflow.modules.CNOFactorized2DBlock": "fourierflow_amd.modules.CNOFactorized2DBlock",
    "fourierflow.modules.CNOFactorizedMesh2D": "fourierflow_amd.modules.CNOFactorizedMesh2D",
    "fourierflow.modules.CNOFactorizedMesh3D": "fourierflow_amd.modules.CNOFactorizedMesh3D",
    "fourierflow.modules.FNOZongyi2DBlock": "fourierflow_amd.modules.FNOZongyi2DBlock",
    "fourierflow.modules.FNOMesh2D": "fourierflow_amd.modules.FNOMesh2D",
    "fourierflow.modules.FNOMesh3D": "fourierflow_amd.modules.FNOMesh3D",
    "fourierflow.modules.WNLinear": "fourierflow_amd.modules.WNLinear",
    "fourierflow.modules.Normalizer": "fourierflow_amd.modules.Normalizer",
    "fourierflow.routines.Grid2DMarkovExperiment": "fourierflow_amd.routines.Grid2DMarkovExperiment",
    "fourierflow.routines.Grid2DRolloutExperiment": "fourierflow_amd.routines.Grid2DRolloutExperiment",
    "fourierflow.routines.StructuredMeshExperiment": "fourierflow_amd.routines.StructuredMeshExperiment",
}
_INTERP = re.compile(r"^\$\{\s*([\w.]+)\s*:\s*(.*?)\s*\}$")


class MethodRef:
    """``${get_method: pkg.mod.attr}`` -- a dotted name, imported lazily (it may not exist here, e.g.
    fourierflow.schedulers.CosineWithWarmupScheduler: only its NAME and kwargs matter to the fused step)."""

    def __init__(self, name: str):
        self.name = name.strip()

    def __repr__(self):
        return f"MethodRef({self.name})"


class Partial:
    """``_target_: functools.partial`` with ``_args_: [method]`` and keyword arguments (config.yaml:36-47)."""

    def __init__(self, func: MethodRef, kwargs: Dict[str, Any]):
        self.func, self.kwargs = func, kwargs

    def __repr__(self):
        return f"Partial({self.func.name}, {self.kwargs})"


def import_string(name: str):
    mod, _, attr = name.rpartition(".")
    return getattr(importlib.import_module(mod), attr)


_INNER = re.compile(r"\$\{([\w.]+):\s*([^${}]*)\}")    # an interpolation with no interpolation inside


def _resolve_scalar(v: Any) -> Any:
    if not isinstance(v, str):
        return v
    # nested interpolations, innermost first: '${eval:2 * ${import:numpy.pi}}' (torus_kochkov configs, `domain`)
    while True:
        inner = None
        for mm in _INNER.finditer(v):
            if not (mm.start() == 0 and mm.end() == len(v.strip())) and mm.group(1) in ("import", "eval"):
                inner = mm
                break
        if inner is None:
            break
        v = v[:inner.start()] + repr(_resolve_scalar(inner.group(0))) + v[inner.end():]
    m = _INTERP.match(v.strip())
    if not m:
        # embedded ${oc.env:VAR} inside a longer string (data paths)
        return re.sub(r"\$\{oc\.env:(\w+)\}", lambda mm: os.environ.get(mm.group(1), ""), v)
    kind, arg = m.group(1), m.group(2)
    if kind == "get_method":
        return MethodRef(arg)
    if kind == "import":
        return import_string(arg)
    if kind == "eval":
        return eval(arg, {"__builtins__": {}}, {})   # arithmetic only, like the reference's resolver use
    if kind == "oc.env":
        return os.environ.get(arg, "")
    raise ValueError(f"unknown resolver ${{{kind}:...}}")


def load_config(path: str, overrides: Sequence[str] = ()) -> Dict[str, Any]:
    """yaml + ``a.b.c=value`` overrides (the positional overrides of `fourierflow train`, train.py:27-34)."""
    with open(path) as f:
        cfg = yaml.safe_load(f)
    for ov in overrides:
        key, _, val = ov.partition("=")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)
    return cfg


def instantiate(node: Any) -> Any:
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if not isinstance(node, dict):
        return _resolve_scalar(node)
    if "_target_" not in node:
        return {k: instantiate(v) for k, v in node.items()}
    target = node["_target_"]
    args = [instantiate(a) for a in node.get("_args_", [])]
    kwargs = {k: instantiate(v) for k, v in node.items() if not k.startswith("_")}
    if target == "functools.partial":
        if not args or not isinstance(args[0], MethodRef):
            raise ValueError("functools.partial nodes need `_args_: [${get_method: ...}]`")
        return Partial(args[0], kwargs)
    mapped = TARGET_MAP.get(target)
    if mapped is None:
        if target.startswith("fourierflow."):
            raise NotImplementedError(f"{target} has no MI355X-native counterpart in fourierflow_amd (see DESIGN.md section 7)")
        mapped = target
    return import_string(mapped)(*args, **kwargs)


def build_routine(cfg: Dict[str, Any]):
    """Instantiate ``cfg['routine']`` (the model + optimiser + schedule); other sections are left alone."""
    r = dict(cfg["routine"])
    opt = instantiate(r.pop("optimizer", None)) if r.get("optimizer") else None
    sch = instantiate(r.pop("scheduler", None)) if r.get("scheduler") else None
    r.pop("optimizer", None)
    r.pop("scheduler", None)
    routine_kwargs = {}
    model_target = str((r.get("conv") or r.get("model") or {}).get("_target_", ""))
    baseline = model_target.endswith(("FNOZongyi2DBlock", "FNOMesh2D", "FNOMesh3D"))     # the StepLR (and Adam) users built here
    if opt is not None:
        names = ("torch.optim.AdamW",) + (("torch.optim.Adam",) if model_target.endswith(("FNOMesh2D", "FNOMesh3D")) else ())
        if not isinstance(opt, Partial) or opt.func.name not in names:
            raise NotImplementedError(f"only torch.optim.AdamW (and torch.optim.Adam for FNOMesh2D / FNOMesh3D) map to the fused flat "
                                      f"optimiser kernel, got {opt}")
        routine_kwargs["optimizer"] = dict(opt.kwargs)
        if opt.func.name == "torch.optim.Adam":
            routine_kwargs["optimizer_type"] = "adam"
    if sch is not None:
        s = sch["scheduler"] if isinstance(sch, dict) else sch
        ok = ("CosineWithWarmupScheduler",) + (("StepLR",) if baseline else ())
        if not isinstance(s, Partial) or not s.func.name.endswith(ok):
            raise NotImplementedError(f"only CosineWithWarmupScheduler (and StepLR for the FNOZongyi2DBlock / FNOMesh2D baselines) are folded into "
                                      f"the fused optimiser step, got {s}")
        routine_kwargs["scheduler"] = dict(s.kwargs)
    node = dict(r)
    target = node.pop("_target_")
    kwargs = {k: instantiate(v) for k, v in node.items() if not k.startswith("_")}
    kwargs.update(routine_kwargs)
    cls = import_string(TARGET_MAP.get(target, target))
    return cls(**kwargs)
