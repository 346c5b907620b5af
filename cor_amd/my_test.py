"""Validation driver: the reference's my_test.py (my_test.py:49-234) on the HIP model.

    python -m cor_amd.my_test --config config/vaild_config.yaml [--soft 1] [--metric 1]

Same YAML keys (batch_size, sam_model_name, siglip_model_name, dataset_path, val_csv_A, val_csv_B, vaild_model_save_path,
mask_pooling, multimask_output, load_checkpoint_path; optional mixed_precision: "bf16" (default) | "no") and the same sequence: config -> logger -> build_model_with_query_support_feat
-> loaders for Test-Base / Test-Novel -> checkpoint ("model_state_dict", optional "module." prefix, strict) -> save_hard_pred_masks
into hard_pred_Test_1 / hard_pred_Test_2. Differences, all forced by the platform: `accelerate` is optional (one process per GPU;
bf16 autocast is applied like config/vaild_config/vaild_a.yaml when no Accelerator is used), the tokenizer comes from a local
sentencepiece file (`siglip_tokenizer_path` in the YAML; a hashing stand-in when absent, with a warning), and the image
transforms run on the GPU (cor_amd/dataloader.py)."""
from __future__ import annotations

import argparse
import logging
import os
from argparse import Namespace
from datetime import datetime

import torch
import yaml

from . import harness, tokenizer
from .dataloader import get_vaild_loader
from .lib.build_model import build_model_with_query_support_feat


def set_seed(seed=0):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def load_config(config_file):
    """ref: my_test.py:31-35."""
    with open(config_file, "r") as f:
        return Namespace(**yaml.safe_load(f))


def init_val_logger(save_path=None, file_name=None):
    """ref: utils/utils.py:109-118."""
    logging.basicConfig(filename=os.path.join(save_path, file_name), format="[%(asctime)s - %(filename)s - %(levelname)s : %(message)s]",
                        level=logging.INFO, filemode="a", datefmt="%Y-%m-%d %I:%M:%S %p", force=True)
    return logging.getLogger()


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Validate a CORE checkpoint on MI355X (HIP engine)")
    ap.add_argument("--config", type=str, required=True, help="Path to the YAML configuration file")
    ap.add_argument("--soft", type=int, default=0, help="1: also save grayscale masks (save_soft_pred_masks)")
    ap.add_argument("--metric", type=int, default=0, help="1: also run val_metric (per-sample CSV + global Dice / MAE / IoU)")
    ap.add_argument("--accelerate", type=int, default=0, help="1: wrap model and loaders with accelerate.Accelerator like the reference")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    opt = load_config(args.config)
    accelerator = None
    if args.accelerate:
        # ref: my_test.py:56 `Accelerator()`: precision comes from the `accelerate launch --config_file` YAML (config/vaild_config/
        # vaild_a.yaml:4 mixed_precision: bf16). Started without that launcher, the YAML key `mixed_precision` of THIS config plays its part.
        from accelerate import Accelerator
        mp = getattr(opt, "mixed_precision", None)
        accelerator = Accelerator(mixed_precision=mp) if mp and "ACCELERATE_MIXED_PRECISION" not in os.environ else Accelerator()
    is_main = accelerator is None or accelerator.is_main_process
    set_seed(0)
    os.makedirs(opt.vaild_model_save_path, exist_ok=True)
    logger = init_val_logger(save_path=opt.vaild_model_save_path, file_name=f"val_log_{datetime.now().strftime('%Y%m%d_%H%M%S')}.log")
    if is_main:
        logger.info(f">>> Training Config: {vars(opt)}")
    if not torch.cuda.is_available():
        raise RuntimeError("cor_amd.my_test needs a GPU: the HIP engine has no CPU path")
    device = accelerator.device if accelerator is not None else torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))

    my_model = build_model_with_query_support_feat(sam_model=opt.sam_model_name, siglip_model=opt.siglip_model_name, sam_checkpoint_path=None,
                                                   siglip_checkpoint_path=None, mask_pooling=opt.mask_pooling)
    spm_path = getattr(opt, "siglip_tokenizer_path", None)
    if spm_path:
        tok = tokenizer.siglip_tokenizer(spm_path)
    else:
        logger.warning("siglip_tokenizer_path not set: using the hashing stand-in tokenizer (NOT the SigLIP vocabulary)")
        tok = tokenizer.hashing_tokenizer(vocab=my_model.support_branch.siglip.cfg["vocab"])
    nw = min((os.cpu_count() or 1), 8)
    loaders = [get_vaild_loader(csv, opt.dataset_path, support_img_size=384, text_tokenizer=tok, batch_size=opt.batch_size, shuffle=False,
                                num_workers=getattr(opt, "num_workers", nw), device=device) for csv in (opt.val_csv_A, opt.val_csv_B)]
    my_model = my_model.to(device).eval()
    if accelerator is not None:
        my_model = accelerator.prepare(my_model)
    if is_main:
        for name, ld in zip("AB", loaders):
            print(f">>> val_loader_{name} with {ld.dataset.dataset_size} samples")
            logger.info(f">>> val_loader_{name} with {ld.dataset.dataset_size} samples")

    if getattr(opt, "load_checkpoint_path", None) is None:
        if is_main:
            logger.warning("No checkpoint to load.")
        raise ValueError("No checkpoint to load.")
    try:
        result, _ = harness.load_core_checkpoint(my_model, opt.load_checkpoint_path, map_location="cpu", strict=True)
        if is_main:
            msg = f">>> Loaded checkpoint from {opt.load_checkpoint_path}."
            logger.info(msg); print(msg)
            if not result.missing_keys and not result.unexpected_keys:
                logger.info("All parameters were correctly loaded and updated.")
    except RuntimeError as e:
        if is_main:
            logger.error(f"Failed to load checkpoint: {e}")
        raise

    try:
        for tag, ld, d in (("Test_A", loaders[0], "Test_1"), ("Test_B", loaders[1], "Test_2")):
            if is_main:
                print(f"Start Valid {tag}..."); logger.info(f"Start Valid {tag}...")
            harness.save_hard_pred_masks(ld, my_model, opt, logger, accelerator=accelerator, dataset_path=opt.dataset_path, pred_save_dir=f"hard_pred_{d}")
            if args.soft:
                harness.save_soft_pred_masks(ld, my_model, opt, logger, accelerator=accelerator, dataset_path=opt.dataset_path, pred_save_dir=f"soft_pred_{d}")
            if args.metric:
                harness.val_metric(ld, my_model, opt, logger, accelerator=accelerator, output_csv_name=f"per_sample_metrics_{d}.csv")
    finally:
        if is_main:
            logger.info(">>> Validation finished!"); print(">>> Validation finished!")
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
