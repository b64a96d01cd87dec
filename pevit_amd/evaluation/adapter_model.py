"""Adapter (post-MLP bottleneck 64) CLIP builder (reference: evaluation/adapter_model.py:285-331,1100-1140)."""
from .model import CLIP, LayerNorm, QuickGELU, VisionTransformer, build_peft_model  # noqa: F401


def build_adapter_model(state_dict: dict):
    return build_peft_model(state_dict, "adapter")
