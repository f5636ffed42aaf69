"""Checkpoint / vocabulary resolution for the engine's loaders (host only, outside the hot path).

The reference downloads weights through open_clip / huggingface_hub / S3 (open_clip_model.py:108-222,
hugging_face_model.py:98-170, core/inference/model_download.py).  That machinery is control plane and out of
scope; this module resolves what is ALREADY on disk:

  * an explicit ``localpath`` (open_clip) or a local directory given as ``name`` (hf),
  * ``$MARQO_AMD_MODEL_DIR/<registry-style name>/`` (e.g. ``open_clip/ViT-B-32/laion2b_s34b_b79k/``,
    ``hf/intfloat/e5-base-v2/``),
  * the Hugging Face hub cache layout (``$HF_HOME/hub/models--org--name/snapshots/*/``).

When nothing is found the loader raises ModelLoadError — unless ``MARQO_AMD_SYNTHETIC_WEIGHTS=1``, in which
case seeded random-init weights of the right architecture are generated (benchmarks / tests: there is no
network for checkpoints).
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, List, Optional

import torch

from marqo_amd.s2_inference.configs import read_env_vars_and_defaults
from marqo_amd.s2_inference.enums import EnvVars

OPEN_CLIP_FILES = ("open_clip_model.safetensors", "open_clip_pytorch_model.bin", "model.safetensors", "pytorch_model.bin")
HF_WEIGHT_FILES = ("model.safetensors", "pytorch_model.bin")


def synthetic_weights_enabled() -> bool:
    return str(read_env_vars_and_defaults(EnvVars.MARQO_AMD_SYNTHETIC_WEIGHTS)).lower() in ("1", "true", "yes")


def model_dir() -> str:
    return str(read_env_vars_and_defaults(EnvVars.MARQO_AMD_MODEL_DIR))


def load_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """.safetensors / torch pickle (.pt, .bin, .pth) -> flat {name: tensor}; unwraps 'state_dict' and 'module.'."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path, device="cpu")
    else:
        obj = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
            obj = obj["state_dict"]
        if not isinstance(obj, dict):
            raise ValueError(f"{path}: expected a state dict, found {type(obj).__name__}")
        sd = obj
    if sd and all(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    return sd


def _first_existing(directory: str, names) -> Optional[str]:
    for n in names:
        p = os.path.join(directory, n)
        if os.path.isfile(p):
            return p
    return None


def _hub_cache_dirs(repo_id: str) -> List[str]:
    roots = []
    for env in ("HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE"):
        if os.environ.get(env):
            roots.append(os.environ[env])
    roots.append(os.path.join(os.environ.get("HF_HOME", os.path.expanduser("~/.cache/huggingface")), "hub"))
    out = []
    for r in roots:
        out.extend(sorted(glob.glob(os.path.join(r, "models--" + repo_id.replace("/", "--"), "snapshots", "*"))))
    return out


def find_open_clip_checkpoint(name: str, localpath: Optional[str]) -> Optional[str]:
    """name: 'open_clip/<arch>/<pretrained>' | 'hf-hub:<repo>' | bare arch (with localpath)."""
    if localpath:
        return localpath if os.path.exists(localpath) else None
    candidates = []
    if name.startswith("hf-hub:"):
        repo = name[len("hf-hub:"):]
        candidates += [os.path.join(model_dir(), "hf-hub", repo)] + _hub_cache_dirs(repo)
    else:
        candidates.append(os.path.join(model_dir(), *name.split("/")))
    for d in candidates:
        if os.path.isfile(d):
            return d
        if os.path.isdir(d):
            f = _first_existing(d, OPEN_CLIP_FILES) or next(iter(sorted(glob.glob(os.path.join(d, "*.pt")))), None)
            if f:
                return f
    return None


def find_hf_dir(name: str) -> Optional[str]:
    """HF repo id or a local directory -> directory holding config.json + weights (+ tokenizer files)."""
    candidates = [name, os.path.join(model_dir(), "hf", *name.split("/")), os.path.join(model_dir(), *name.split("/"))]
    candidates += _hub_cache_dirs(name)
    for d in candidates:
        if os.path.isdir(d) and os.path.isfile(os.path.join(d, "config.json")) and _first_existing(d, HF_WEIGHT_FILES):
            return d
    return None


def load_hf_dir(directory: str):
    with open(os.path.join(directory, "config.json")) as f:
        cfg = json.load(f)
    sd = load_state_dict(_first_existing(directory, HF_WEIGHT_FILES))
    return cfg, sd


def read_pooling_config(directory: Optional[str]) -> Optional[str]:
    """sentence-transformers '1_Pooling/config.json' -> 'cls' | 'mean' | None
    (reference: hugging_face_model_properties.py:88-124 reads the same file from the hub)."""
    if not directory:
        return None
    p = os.path.join(directory, "1_Pooling", "config.json")
    try:
        with open(p) as f:
            content = json.load(f)
    except (OSError, json.JSONDecodeError):
        return None
    if not isinstance(content, dict):
        return None
    if content.get("pooling_mode_cls_token") is True:
        return "cls"
    if content.get("pooling_mode_mean_tokens") is True:
        return "mean"
    return None
