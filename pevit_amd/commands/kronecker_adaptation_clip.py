"""python -m pevit_amd.commands.kronecker_adaptation_clip --ds <dataset.yaml> --model <model.yaml> [--no-tuning True --lr .. --l2 ..] [KEY VALUE ...]
(reference: commands/kronecker_adaptation_clip.py)."""
from ..evaluation.kadaptation_clip import kadapt_clip
from ._finetune import run


def main(argv=None):
    return run(kadapt_clip, argv)


if __name__ == "__main__":
    main()
