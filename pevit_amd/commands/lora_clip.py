"""python -m pevit_amd.commands.lora_clip --ds <dataset.yaml> --model <model.yaml> [--no-tuning True --lr .. --l2 ..] [KEY VALUE ...]
(reference: commands/lora_clip.py)."""
from ..evaluation.lora_clip import lora_tuning_clip
from ._finetune import run


def main(argv=None):
    return run(lora_tuning_clip, argv)


if __name__ == "__main__":
    main()
