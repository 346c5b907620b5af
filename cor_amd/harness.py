"""Inference-harness counterpart of the reference's utils/vailder.py for the HIP model.

Same entry points, arguments and file outputs as the reference:
  * save_hard_pred_masks(test_loader, model, opt, logger, accelerator=None, dataset_path=..., pred_save_dir=...)
        utils/vailder.py:368-510 — sigmoid -> per-sample min-max -> resize to the ground-truth size -> > 0.5 -> PNG
  * save_soft_pred_masks(...)   utils/vailder.py:513-656 — the same with grayscale output ((pred * 255).astype(uint8))
  * val_metric(test_loader, model, opt, logger, accelerator=None, output_csv_name=...)
        utils/vailder.py:13-221 (commented out in the shipped reference, restored here) with the metric definitions of
        utils/trainer_v3_g.py:381-443 — per-sample CSV + global Dice / MAE / IoU / mDice / mIoU
The device-side arithmetic (sigmoid + min-max, bilinear resize + threshold, metrics) runs in HIP kernels
(cor_amd/csrc/postproc.hip); cv2 and accelerate are NOT required (cv2.resize(INTER_LINEAR) == the half-pixel bilinear
kernel; `accelerator` may be None for a single process, in which case bf16 autocast is applied like vaild_a.yaml:4).
Batches follow the reference's loader contract (utils/dataloader.py:244-369): dict with query_img, support_img,
support_mask, text, pair_id, query_mask_name, dataset, target (+ query_mask, compose, query_cat for val_metric).
"""
from __future__ import annotations

import contextlib
import csv
import functools
import os
import time
from datetime import datetime, timedelta

import torch

from . import ops


class AverageMeter:
    """ref: utils/utils.py (running sum / count / average)."""

    def __init__(self):
        self.total_sum, self.count = 0.0, 0

    def update(self, val, n=1):
        self.total_sum += float(val) * n
        self.count += n

    @property
    def average(self):
        return self.total_sum / max(self.count, 1)


def _is_main(accelerator):
    return True if accelerator is None else bool(accelerator.is_main_process)


def _autocast(accelerator, opt=None):
    """accelerator.autocast() like the reference; without accelerate the precision comes from the config key `mixed_precision`
    ("bf16" = config/vaild_config/vaild_a.yaml:4, the default; "no" = the exact fp32 MFMA mode)."""
    if accelerator is not None:
        return accelerator.autocast()
    mp = getattr(opt, "mixed_precision", "bf16") if opt is not None else "bf16"
    if mp not in ("bf16", "no"):
        raise ValueError(f"mixed_precision must be 'bf16' or 'no' (fp16 autocast has no kernels here), got {mp!r}")
    return torch.autocast("cuda", dtype=torch.bfloat16) if (mp == "bf16" and torch.cuda.is_available()) else contextlib.nullcontext()


def _dev(model):
    return getattr(model, "device", None) or next(model.parameters()).device


def _on_model_device(fn):
    """Run a harness loop with the model's GPU as the current device (the HIP kernels are enqueued on the current device)."""
    @functools.wraps(fn)
    def wrapped(test_loader, model, *a, **k):
        dev = _dev(model)
        if dev.type != "cuda":
            raise RuntimeError("cor_amd harness: the model must live on a GPU (there is no CPU path)")
        with torch.cuda.device(dev):
            return fn(test_loader, model, *a, **k)
    return wrapped


def postprocess_masks(pred_mask: torch.Tensor) -> torch.Tensor:
    """sigmoid -> per-sample min-max. ref: utils/vailder.py:426-430. [B,1,H,W] logits -> [B,1,H,W] in [0,1]."""
    return ops.mask_prob_minmax(pred_mask)


def compute_metrics(pred: torch.Tensor, gt: torch.Tensor, smooth: float = 1e-5) -> torch.Tensor:
    """[B,5] = dice, mae, iou, mdice, miou per sample. ref: utils/trainer_v3_g.py:381-443."""
    return ops.mask_metrics(pred, gt, smooth)


def _save_pred_masks(test_loader, model, opt, logger, accelerator, dataset_path, pred_save_dir, soft):
    from PIL import Image
    model.eval()
    main = _is_main(accelerator)
    if main:
        print("=" * 35, "Save model predictions", "=" * 35)
    pred_save_path = os.path.join(opt.vaild_model_save_path, pred_save_dir)
    if main:
        os.makedirs(pred_save_path, exist_ok=True)
        print(f"[INFO] Prediction masks will be saved to: {pred_save_path}")
    meter, t_epoch, total = AverageMeter(), time.time(), len(test_loader)
    dev = _dev(model)
    for batch_idx, batch in enumerate(test_loader, start=1):
        t0 = time.time()
        with _autocast(accelerator, opt):
            pred_mask, _, _ = model(query_image_inputs=batch["query_img"].to(dev), support_image_inputs=batch["support_img"].to(dev),
                                    change_text_inputs=batch["text"].to(dev), support_mask_inputs=batch["support_mask"].to(dev),
                                    multimask_output=opt.multimask_output)
        pred = postprocess_masks(pred_mask)                                   # [B,1,256,256] on the GPU
        if main:
            for i in range(pred.shape[0]):
                name = batch["query_mask_name"][i]
                gt_path = os.path.join(dataset_path, str(batch["dataset"][i]), "mask", str(batch["target"][i]), name)
                if not os.path.exists(gt_path):
                    logger.warning(f"GT mask not found: {gt_path}, skipping sample")
                    continue
                try:
                    with Image.open(gt_path) as gt:
                        gw, gh = gt.size
                except Exception as e:                                        # noqa: BLE001 (reference logs and skips)
                    logger.error(f"Failed to read GT mask size for {gt_path}: {e}")
                    continue
                if gw <= 1 or gh <= 1:
                    logger.error(f"Invalid GT size {(gw, gh)} for {gt_path}, skipping sample")
                    continue
                one = pred[i:i + 1].contiguous()
                img = (ops.resize_gray(one, gh, gw) if soft else ops.resize_binarize(one, gh, gw, 0.5))[0].cpu().numpy()
                out_path = os.path.join(pred_save_path, f"{_item(batch['pair_id'][i])}_{name}")
                try:
                    Image.fromarray(img).save(out_path)
                except Exception as e:                                        # noqa: BLE001
                    logger.error(f"Failed to save prediction mask {out_path}: {e}")
        meter.update(time.time() - t0)
        if main and (batch_idx % 10 == 0 or batch_idx == total):
            eta = str(timedelta(seconds=int(meter.average * (total - batch_idx))))
            print(f"{datetime.now().strftime('%Y-%m-%d %H:%M:%S')} [Batch: {batch_idx:04d}/{total:04d}] => [ETA: {eta}]")
    dur = str(timedelta(seconds=int(time.time() - t_epoch)))
    if main:
        logger.info(f"Predictions saved to {pred_save_path}, [Duration: {dur}]")
        print(f"Predictions saved to {pred_save_path}, [Duration: {dur}]")


@_on_model_device
@torch.no_grad()
def save_hard_pred_masks(test_loader, model, opt, logger, accelerator=None, dataset_path="/data/dataset", pred_save_dir="predictions"):
    """ref: utils/vailder.py:368-510 - sigmoid -> min-max -> resize to the GT size -> > 0.5 -> uint8 * 255 -> PNG."""
    _save_pred_masks(test_loader, model, opt, logger, accelerator, dataset_path, pred_save_dir, soft=False)


@_on_model_device
@torch.no_grad()
def save_soft_pred_masks(test_loader, model, opt, logger, accelerator=None, dataset_path="/data/dataset", pred_save_dir="predictions"):
    """ref: utils/vailder.py:513-656 - as save_hard_pred_masks, but the resized probabilities are kept as grayscale:
    (pred * 255).astype(np.uint8) instead of the 0.5 threshold."""
    _save_pred_masks(test_loader, model, opt, logger, accelerator, dataset_path, pred_save_dir, soft=True)


def _item(v):
    return v.item() if torch.is_tensor(v) else v


_FIELDS = ["Id", "Query_img", "Query_mask", "Support_img", "Support_mask", "Text", "Compose", "Dataset", "Target", "query_cat",
           "Dice", "MAE", "IoU", "mDice", "mIoU"]


@_on_model_device
@torch.no_grad()
def val_metric(test_loader, model, opt, logger, accelerator=None, output_csv_name="per_sample_metrics.csv"):
    model.eval()
    main = _is_main(accelerator)
    if main:
        print("=" * 35, "Valid model", "=" * 35)
        os.makedirs(opt.vaild_model_save_path, exist_ok=True)
    csv_path = os.path.join(opt.vaild_model_save_path, output_csv_name)
    if main:
        with open(csv_path, "w", newline="") as f:
            csv.DictWriter(f, fieldnames=_FIELDS).writeheader()
    sums, n_samples = torch.zeros(5, dtype=torch.float64), 0
    dev = _dev(model)
    t_epoch = time.time()
    for batch in test_loader:
        gt = batch["query_mask"].to(dev).float()
        with _autocast(accelerator, opt):
            pred_mask, _, _ = model(query_image_inputs=batch["query_img"].to(dev), support_image_inputs=batch["support_img"].to(dev),
                                    change_text_inputs=batch["text"].to(dev), support_mask_inputs=batch["support_mask"].to(dev),
                                    multimask_output=opt.multimask_output)
        if tuple(pred_mask.shape[-2:]) != tuple(gt.shape[-2:]):              # F.interpolate(..., bilinear, align_corners=False)
            pred_mask = ops.bilinear(pred_mask.float().contiguous(), gt.shape[-2], gt.shape[-1])
        pred = postprocess_masks(pred_mask)
        m = compute_metrics(pred, gt.contiguous()).cpu()                      # [B,5]
        sums += m.double().sum(0)
        n_samples += m.shape[0]
        if main:
            with open(csv_path, "a", newline="") as f:
                w = csv.DictWriter(f, fieldnames=_FIELDS)
                for i in range(m.shape[0]):
                    pid = _item(batch["pair_id"][i])
                    get = lambda k, d: (batch[k][i] if batch.get(k) is not None else d)   # noqa: E731
                    w.writerow({"Id": pid, "Query_img": get("query_img_name", f"sample_{pid}_query_img"),
                                "Query_mask": get("query_mask_name", f"sample_{pid}_query_mask"),
                                "Support_img": get("support_img_name", f"sample_{pid}_support_img"),
                                "Support_mask": get("support_mask_name", f"sample_{pid}_support_mask"),
                                "Text": get("text_string", ""), "Compose": _item(get("compose", "")), "Dataset": get("dataset", ""),
                                "Target": get("target", ""), "query_cat": _item(get("query_cat", "")),
                                "Dice": f"{m[i, 0]:.4f}", "MAE": f"{m[i, 1]:.4f}", "IoU": f"{m[i, 2]:.4f}",
                                "mDice": f"{m[i, 3]:.4f}", "mIoU": f"{m[i, 4]:.4f}"})
    if accelerator is not None:                                               # weighted global average over processes
        accelerator.wait_for_everyone()
        t = torch.cat([sums, torch.tensor([float(n_samples)], dtype=torch.float64)]).to(accelerator.device)
        t = accelerator.gather(t.unsqueeze(0)).sum(0).cpu()
        sums, n_samples = t[:5], int(t[5].item())
    g = (sums / max(n_samples, 1)).tolist()
    dur = str(timedelta(seconds=int(time.time() - t_epoch)))
    if main:
        msg = (f"Global Dice: {g[0]:.4f}, Global MAE: {g[1]:.4f}, Global IoU: {g[2]:.4f}, Global mDice: {g[3]:.4f}, "
               f"Global mIoU: {g[4]:.4f}, [Duration: {dur}]")
        logger.info(msg)
        print(msg)
        logger.info(f"Per-sample metrics saved to {csv_path}")
    return {"global_metrics": {"dice": g[0], "mae": g[1], "iou": g[2], "mdice": g[3], "miou": g[4]}, "per_sample_metrics": None}


def load_core_checkpoint(model, path, map_location="cpu", strict=True):
    """Load a released CORE checkpoint the way my_test.py:118-145 does: the weights live under "model_state_dict"
    (utils/trainer_v3_g.py:134-142), and keys carry a "module." prefix when they were saved from a DDP-wrapped model."""
    ckpt = torch.load(path, map_location=map_location)
    sd = ckpt.get("model_state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
    if any(k.startswith("module.") for k in sd):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    result = model.load_state_dict(sd, strict=strict)
    return result, (ckpt.get("epoch") if isinstance(ckpt, dict) else None)
