"""python -m pevit_amd.commands.adapter_clip --ds <dataset.yaml> --model <model.yaml> [--no-tuning True --lr .. --l2 ..] [KEY VALUE ...]
(reference: commands/adapter_clip.py)."""
from ..evaluation.adapter_tuning_clip import adapt_clip
from ._finetune import run


def main(argv=None):
    return run(adapt_clip, argv)


if __name__ == "__main__":
    main()
