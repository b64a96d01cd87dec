"""Compacter (PHM n=4 bottleneck 64) CLIP builder (reference: evaluation/compacter_model.py:398-520,1290-1330)."""
from .model import CLIP, LayerNorm, QuickGELU, VisionTransformer, build_peft_model  # noqa: F401


def build_compacter_model(state_dict: dict):
    return build_peft_model(state_dict, "compacter")
