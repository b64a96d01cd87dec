"""python -m pevit_amd.commands.compacter_clip --ds <dataset.yaml> --model <model.yaml> [--no-tuning True --lr .. --l2 ..] [KEY VALUE ...]
(reference: commands/compacter_clip.py)."""
from ..evaluation.compacter_clip import compacter_clip
from ._finetune import run


def main(argv=None):
    return run(compacter_clip, argv)


if __name__ == "__main__":
    main()
