"""Host-side mirror of ``vision_benchmark.evaluation`` for the PEFT fine-tune path.

Same importable names and call signatures as the reference's Python modules; the vision tower,
head, loss and optimiser execute in ``libpevit_hip.so``.  Dataset download/manifest code, the BPE
vocabulary and the knowledge-augmented prompt sources of the reference are not part of this path.
"""
from .dataloader import construct_dataloader  # noqa: E402,F401
from .feature import extract_text_features  # noqa: E402,F401
