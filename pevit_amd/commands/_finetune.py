"""Command-line entry shared by the four fine-tune commands (reference: commands/kronecker_adaptation_clip.py:28-173
and its lora / adapter / compacter twins): same flags, same config layering (dataset yaml <- model yaml <- KEY VALUE
opts), same outputs -- ``<OUTPUT_DIR>/predictions/<exp>/seed<S>_<dataset>.json`` with the leaderboard keys and the
one-line ``.txt`` summary that read_txt.py / read_results.py parse."""
from __future__ import annotations

import argparse
import json
import logging
import os
import random

import numpy as np
import torch

from ..config import config, update_config
from ..evaluation.dataloader import construct_dataloader


def add_finetuning_args(parser):
    parser.add_argument("--ds", required=False, help="Evaluation dataset configure file name.", type=str)
    parser.add_argument("--model", required=True, help="Evaluation model configure file name", type=str)
    parser.add_argument("--submit-predictions", help="submit predictions and model info to leaderboard.", default=False,
                        action="store_true")
    parser.add_argument("--submit-by", help="Person who submits the results.", type=str)
    parser.add_argument("--no-tuning", help="No hyperparameter-tuning.", default=False, type=lambda x: x.lower() == "true")
    parser.add_argument("--l2", help="(Inverse) L2 regularization strength; used when --no-tuning is True.", default=0.316, type=float)
    parser.add_argument("--lr", help="Learning rate; used when --no-tuning is True.", default=0.001, type=float)
    parser.add_argument("--run", help="Run id", default=1, type=int)
    parser.add_argument("--fix_seed", help="Fix the random seed. [-1] not fixing the seeds", default=0, type=int)
    parser.add_argument("--save-predictions", help="save predictions logits for analysis.", default=True, action="store_true")
    parser.add_argument("opts", help="Modify config options using the command-line", default=None, nargs=argparse.REMAINDER)


def json_prec_dump(data, prec=6):
    return json.dumps(json.loads(json.dumps(data), parse_float=lambda x: round(float(x), prec)))


def write_results(cfg, exp_name, best_acc, model_info):
    preds = model_info["best_logits"]
    results = {
        "model_name": cfg.MODEL.NAME, "dataset_name": cfg.DATASET.DATASET,
        "num_trainable_params": model_info.get("n_trainable_params"), "num_params": model_info.get("n_params"),
        "num_visual_params": model_info.get("n_visual_params"), "num_backbone_params": model_info.get("n_backbone_params"),
        "n_shot": cfg.DATASET.NUM_SAMPLES_PER_CLASS, "rnd_seeds": [cfg.DATASET.RANDOM_SEED_SAMPLING],
        "predictions": [np.asarray(preds).tolist()],
    }
    folder = os.path.join(cfg.OUTPUT_DIR, "predictions", exp_name)
    os.makedirs(folder, exist_ok=True)
    stem = os.path.join(folder, f"seed{cfg.DATASET.RANDOM_SEED_SAMPLING}_{cfg.DATASET.DATASET}")
    with open(stem + ".json", "w") as f:
        f.write(json_prec_dump(results))
    with open(stem + ".txt", "w") as f:
        f.write(f"best acc is:{best_acc}, num_params is:{model_info.get('n_params')}, "
                f"n_trainable_params is:{model_info.get('n_trainable_params') / 1000000}, "
                f"backbone_params is:{model_info.get('n_backbone_params')}.")
    return stem


def run(entry, argv=None, description="Test a classification model, with finetuning."):
    parser = argparse.ArgumentParser(description=description)
    add_finetuning_args(parser)
    args = parser.parse_args(argv)
    opts = args.opts
    if args.ds:
        args.cfg = args.ds
        update_config(config, args)
    args.cfg = args.model
    update_config(config, args)
    config.defrost()
    config.NAME = ""
    config.freeze()
    if args.submit_predictions:
        raise SystemExit("--submit-predictions talks to the reference's leaderboard service; not available here")
    if args.fix_seed != -1:
        random.seed(args.fix_seed); np.random.seed(args.fix_seed)
        torch.manual_seed(args.fix_seed); torch.cuda.manual_seed_all(args.fix_seed)
    n_samples = str(config.DATASET.NUM_SAMPLES_PER_CLASS) if config.DATASET.NUM_SAMPLES_PER_CLASS > 0 else "full"
    exp_name = "finetuning_" + n_samples + ("_two_lr" if config.TRAIN.TWO_LR else "")
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    if config.DATASET.NUM_SAMPLES_PER_CLASS == 1:
        config.defrost()
        config.DATASET.NUM_SAMPLES_PER_CLASS = 2
        config.DATASET.MERGE_TRAIN_VAL_FINAL_RUN = False
        config.freeze()
    logging.info(f"{config.DATASET.DATASET} is a dataset.")
    train_dl, val_dl, test_dl = construct_dataloader(config)
    logging.info("Finetuning with full model. This may take several minutes to hours depending on the size of your data.")
    best_acc, model_info = entry(train_dl, val_dl, test_dl, args.no_tuning, args.lr, args.l2, config)
    if args.save_predictions:
        stem = write_results(config, exp_name, best_acc, model_info)
        logging.info(f"results written to {stem}.json / .txt")
    return best_acc, model_info
