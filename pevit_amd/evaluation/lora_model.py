"""LoRA CLIP builder (reference: evaluation/lora_model.py:1142-1182; r=4, alpha=128 hard-coded at :461-463)."""
from .model import CLIP, LayerNorm, QuickGELU, VisionTransformer, build_peft_model  # noqa: F401


def build_lora_model(state_dict: dict, lora_rank: int = 4):
    return build_peft_model(state_dict, "lora", lora_rank)
