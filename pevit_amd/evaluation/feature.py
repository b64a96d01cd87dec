"""Zero-shot head initialisation and loader helpers (reference: evaluation/feature.py:405-531,585-608).

Only what ``Classifier(INIT_HEAD_WITH_TEXT_ENCODER=True)`` needs is here: prompts per class -> text tower
(host PyTorch, run once) -> normalise / mean over templates / normalise -> (D, C) weight matrix.  The
reference's class-name tables, prompt templates, tokenizer vocabulary and knowledge sources are data of
its dataset layer and are not shipped: the caller supplies them through ``config.DATASET.CLASS_NAMES`` /
``config.DATASET.TEMPLATES`` (or the keyword arguments) and a ``tokenizer(texts, context_length=...)``.
"""
from __future__ import annotations

import logging
import time

import torch


@torch.no_grad()
def extract_text_features(config, tokenizer, args=None, model=None, return_numpy=True, class_names=None, templates=None):
    if model is None:
        raise RuntimeError("extract_text_features needs the CLIP model whose text tower is to be used")
    class_names = class_names or config.DATASET.get("CLASS_NAMES", None)
    if not class_names:
        raise RuntimeError("no class names: set config.DATASET.CLASS_NAMES (the reference's class_map / dataset-hub "
                           "lookup, feature.py:408-415, belongs to its dataset layer)")
    templates = templates or config.DATASET.get("TEMPLATES", None) or ["a photo of a {}"]
    know = config.get("KNOWLEDGE", None)
    if know is not None and (know.WIKITIONARY.USE_DEFINITION or know.WORDNET.USE_DEFINITION or know.WORDNET.USE_HIERARCHY
                             or know.GPT3.USE_GPT3):
        raise RuntimeError("knowledge-augmented prompts (KNOWLEDGE.*) are not part of this build")
    device = next(model.parameters()).device
    start = time.time()
    model.eval()
    cols = []
    ctx_len = config.MODEL.SPEC.TEXT.CONTEXT_LENGTH
    for classname in class_names:
        if type(classname) == list:
            classname = classname[0]
        if torch.is_tensor(classname):                    # already tokenised prompts of this class: (T, context)
            texts = classname.to(device)
        else:
            texts = [t.format(classname) for t in templates]
            if not config.MODEL.SPEC.TEXT.get("SKIP_TOKENIZE", False):
                if tokenizer is None:
                    raise RuntimeError("a tokenizer callable is required to turn prompts into token ids")
                texts = tokenizer(texts, context_length=ctx_len).to(device)
        emb = model.encode_text(texts)
        emb = emb / emb.norm(dim=-1, keepdim=True)
        emb = emb.mean(dim=0)
        cols.append(emb / emb.norm())
    zeroshot_weights = torch.stack(cols, dim=1).to(device)
    logging.info(f"=> Feature extraction duration time: {time.time() - start:.2f}s")
    return zeroshot_weights.cpu().detach().numpy() if return_numpy else zeroshot_weights


def create_dataloader(dataset, batch_size, shuffle=True, num_workers=6, pin_memory=True):
    """feature.py:585-608: a plain DataLoader without sampler / drop_last."""
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers,
                                       pin_memory=pin_memory, sampler=None, drop_last=False)
