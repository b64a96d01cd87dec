"""Checkpoint loading with the reference's ``clip_load`` surface (evaluation/clip_load.py:95-190 and the
``adapter_load`` / ``lora_load`` / ``compacter_load`` siblings at :193,:290,:387).

``name`` is a key of ``_MODELS`` or a path to a checkpoint (TorchScript archive or plain state-dict).
There is no network access in the target environment, so a ``_MODELS`` key is resolved against
``download_root`` (default ``~/.cache/clip``) and its SHA-256 is verified like clip_load.py:52-71; a
missing file or a bad digest is a ``RuntimeError`` exactly as in the reference.  The returned model is
the engine-backed mirror in :mod:`.model`; the JIT execution mode of the reference (``jit=True``) has
no meaning for it and is rejected.
"""
from __future__ import annotations

import hashlib
import os
import warnings
from typing import List, Union

import numpy as np
import torch

from .adapter_model import build_adapter_model
from .compacter_model import build_compacter_model
from .lora_model import build_lora_model
from .model import build_model

__all__ = ["available_models", "load", "adapter_load", "lora_load", "compacter_load"]

# name -> (file name, sha256) of the OpenAI release the reference downloads (clip_load.py:32-41)
_MODELS = {
    "RN50": ("RN50.pt", "afeb0e10f9e5a86da6080e35cf09123aca3b358a0c3e3b6c78a7b63bc04b6762"),
    "RN101": ("RN101.pt", "8fa8567bab74a42d41c5915025a8e4538c3bdbe8804a470a72f30b0d94fab599"),
    "RN50x4": ("RN50x4.pt", "7e526bd135e493cef0776de27d5f42653e6b4c8bf9e0f653bb11773263205fdd"),
    "RN50x16": ("RN50x16.pt", "52378b407f34354e150460fe41077663dd5b39c54cd0bfd2b27167a4a06ec9aa"),
    "RN50x64": ("RN50x64.pt", "be1cfb55d75a9666199fb2206c106743da0f6468c9d327f3e0d0a543a9919d9c"),
    "ViT-B/32": ("ViT-B-32.pt", "40d365715913c9da98579312b702a82c18be219cc2a73407c4526f58eba950af"),
    "ViT-B/16": ("ViT-B-16.pt", "5806e77cd80f8b59890b7e101eabd078d9fb84e6937f9e85e4ecb61988df416f"),
    "ViT-L/14": ("ViT-L-14.pt", "b8cca3fd41ae0c99ba7e8951adf17d267cdb84cd88be6f7c2e0eca1737a03836"),
}

_DEFAULT_DEVICE = "cuda" if torch.cuda.is_available() else "cpu"
_SD_CACHE: dict = {}          # (path, mtime, size) -> CPU state-dict; the sweep reloads the same file ~90 times


def _resolve(filename: str, expected_sha256: str, root: str) -> str:
    target = os.path.join(root, filename)
    if os.path.exists(target) and not os.path.isfile(target):
        raise RuntimeError(f"{target} exists and is not a regular file")
    if not os.path.isfile(target):
        raise RuntimeError(f"{target} not found and this build has no network access: place the OpenAI "
                           f"checkpoint there or pass a file path as `name`")
    h = hashlib.sha256()
    with open(target, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    if h.hexdigest() != expected_sha256:
        raise RuntimeError("Model file is present but the SHA256 checksum does not not match")
    return target


class _Preprocess:
    """Resize(n_px, bicubic) -> CenterCrop -> RGB -> [0,1] tensor -> Normalize(CLIP mean/std)
    (clip_load.py:80-87), on PIL images, without torchvision."""
    MEAN = (0.48145466, 0.4578275, 0.40821073)
    STD = (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, n_px: int):
        self.n_px = n_px

    def __call__(self, image):
        from PIL import Image
        w, h = image.size
        n = self.n_px
        if w <= h:
            nw, nh = n, int(n * h / w)
        else:
            nw, nh = int(n * w / h), n
        if (nw, nh) != (w, h):
            image = image.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - n) / 2.0)), int(round((nh - n) / 2.0))
        image = image.crop((left, top, left + n, top + n)).convert("RGB")
        x = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
        mean = torch.tensor(self.MEAN).view(3, 1, 1)
        std = torch.tensor(self.STD).view(3, 1, 1)
        return (x - mean) / std

    def __repr__(self):
        return f"_Preprocess(n_px={self.n_px})"


def _transform(n_px):
    return _Preprocess(n_px)


def available_models() -> List[str]:
    """Returns the names of available CLIP models"""
    return list(_MODELS.keys())


def _read_state_dict(model_path: str):
    st = os.stat(model_path)
    key = (os.path.abspath(model_path), st.st_mtime_ns, st.st_size)
    if key in _SD_CACHE:
        return _SD_CACHE[key]
    try:
        sd = torch.jit.load(model_path, map_location="cpu").eval().state_dict()      # OpenAI releases are JIT archives
    except RuntimeError:
        sd = torch.load(model_path, map_location="cpu", weights_only=True)      # tensors only: no pickled code
    if isinstance(sd, dict) and "state_dict" in sd and "visual.proj" not in sd:
        sd = sd["state_dict"]
    _SD_CACHE.clear()
    _SD_CACHE[key] = sd
    return sd


def _load(builder, name, device, jit, download_root):
    if name in _MODELS:
        model_path = _resolve(*_MODELS[name], download_root or os.path.expanduser("~/.cache/clip"))
    elif os.path.isfile(name):
        model_path = name
    else:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    if jit:
        raise RuntimeError("jit=True is not supported: the vision tower executes in the HIP engine, not TorchScript")
    state_dict = _read_state_dict(model_path)
    model = builder(dict(state_dict)).to(device)
    if str(device) == "cpu":
        model.float()
    return model, _transform(model.visual.input_resolution)


def load(name: str, device: Union[str, torch.device] = _DEFAULT_DEVICE, jit: bool = False, download_root: str = None):
    """KAdaptation CLIP + preprocess (clip_load.py:95)."""
    return _load(build_model, name, device, jit, download_root)


def adapter_load(name: str, device: Union[str, torch.device] = _DEFAULT_DEVICE, jit: bool = False, download_root: str = None):
    return _load(build_adapter_model, name, device, jit, download_root)


def lora_load(name: str, device: Union[str, torch.device] = _DEFAULT_DEVICE, jit: bool = False, download_root: str = None):
    return _load(build_lora_model, name, device, jit, download_root)


def compacter_load(name: str, device: Union[str, torch.device] = _DEFAULT_DEVICE, jit: bool = False, download_root: str = None):
    return _load(build_compacter_model, name, device, jit, download_root)
