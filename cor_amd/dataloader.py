"""Validation data path of the reference on the HIP pre-processing kernels.

Mirrors utils/dataloader.py:244-413 (class VaildingDataset, get_vaild_loader): same CSV schema
(Id, Query_img, Query_mask, Support_img, Support_mask, Text, Compose, Dataset, Target, query_cat; rows with Compose != 0 are
dropped), same directory layout (<dataset_path>/<Dataset>/image/<file>, .../mask/<Target>/<Query_mask>, .../mask/sup/
<Support_mask>), same batch dict keys. MI355X-first split of the work: DataLoader workers only DECODE files (PIL, host);
`Resize -> ToTensor -> Normalize` runs on the GPU in the collate step of the main process (cor_amd/preprocess.py: Pillow's
antialiased bilinear resize bit-exactly, csrc/preproc.hip), so a batch arrives on the device already at model size.
The SigLIP tokenizer is resolved by name through open_clip in the reference (an HF-hub download): here `text_tokenizer` is a
callable str -> LongTensor[64] (cor_amd.tokenizer.siglip_tokenizer builds one from a local sentencepiece model file)."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import preprocess as P

CSV_COLUMNS = ["Id", "Query_img", "Query_mask", "Support_img", "Support_mask", "Text", "Compose", "Dataset", "Target", "query_cat"]


def read_pairs_csv(csv_path):
    """ref: utils/dataloader.py:258-261 - pandas frame of the pairs with Compose == 0."""
    import pandas as pd
    df = pd.read_csv(csv_path)
    missing = [c for c in CSV_COLUMNS if c not in df.columns]
    if missing:
        raise ValueError(f"{csv_path}: missing CSV columns {missing}")
    return df[df["Compose"] == 0].reset_index(drop=True)


class VaildingDataset(Dataset):
    """ref: utils/dataloader.py:244-369. __getitem__ returns DECODED uint8 arrays (no resize: that happens on the GPU in
    `collate_on_device`) plus the reference's metadata fields."""

    def __init__(self, csv_path, dataset_path, support_img_size=384, text_tokenizer=None):
        self.dataset_csv = read_pairs_csv(csv_path)
        self.dataset_path = dataset_path
        self.query_img_size = 1024                     # "use sam pretrain can't change" (dataloader.py:262)
        self.support_img_size = support_img_size
        self.support_mask_size = support_img_size
        if text_tokenizer is None or isinstance(text_tokenizer, str):
            raise ValueError("text_tokenizer must be a callable str -> LongTensor[ctx] (the reference resolves a hub tokenizer by "
                             "name, which needs network access); see cor_amd.tokenizer")
        self.siglip_text_tokenizer = text_tokenizer
        self.dataset_size = len(self.dataset_csv)

    @staticmethod
    def rgb_loader(path):
        from PIL import Image
        with open(path, "rb") as f:
            return np.asarray(Image.open(f).convert("RGB"))

    @staticmethod
    def binary_loader(path):
        from PIL import Image
        with open(path, "rb") as f:
            return np.asarray(Image.open(f).convert("L"))

    def __len__(self):
        return self.dataset_size

    def __getitem__(self, idx):
        row = self.dataset_csv.iloc[idx]
        ds, tgt = row["Dataset"], row["Target"]
        root = os.path.join(self.dataset_path, ds)
        text_string = row["Text"]
        tokens = torch.as_tensor(self.siglip_text_tokenizer(text_string)).reshape(-1).to(torch.int64)
        return {
            "pair_id": row["Id"],
            "query_img_u8": self.rgb_loader(os.path.join(root, "image", row["Query_img"])),
            "query_mask_u8": self.binary_loader(os.path.join(root, "mask", str(tgt), row["Query_mask"])),
            "support_img_u8": self.rgb_loader(os.path.join(root, "image", row["Support_img"])),
            "support_mask_u8": self.binary_loader(os.path.join(root, "mask", "sup", row["Support_mask"])),
            "text": tokens, "text_string": text_string, "compose": int(row["Compose"]), "dataset": ds, "target": tgt,
            "query_cat": row["query_cat"], "query_img_name": row["Query_img"], "query_mask_name": row["Query_mask"],
            "support_img_name": row["Support_img"], "support_mask_name": row["Support_mask"],
        }


class collate_on_device:
    """Stacks a list of samples into the reference's batch dict; the four image fields are resized / normalised on `device`
    (utils/dataloader.py:266-293: Resize -> ToTensor -> Normalize for images, Resize -> ToTensor for masks)."""

    def __init__(self, device, support_img_size=384, query_img_size=1024):
        self.device = torch.device(device)
        self.q_img, self.q_mask = P.ImageTransform(query_img_size), P.MaskTransform(query_img_size)
        self.s_img, self.s_mask = P.ImageTransform(support_img_size), P.MaskTransform(support_img_size)

    def __call__(self, samples):
        dev = self.device
        up = lambda a: torch.from_numpy(np.array(a, copy=True)).to(dev)               # noqa: E731 (PIL hands out read-only arrays)
        out = {
            "query_img": torch.stack([self.q_img(up(s["query_img_u8"])) for s in samples]),
            "query_mask": torch.stack([self.q_mask(up(s["query_mask_u8"])) for s in samples]),
            "support_img": torch.stack([self.s_img(up(s["support_img_u8"])) for s in samples]),
            "support_mask": torch.stack([self.s_mask(up(s["support_mask_u8"])) for s in samples]),
            "text": torch.stack([s["text"] for s in samples]).to(dev),
        }
        for k in ("pair_id", "text_string", "compose", "dataset", "target", "query_cat", "query_img_name", "query_mask_name",
                  "support_img_name", "support_mask_name"):
            out[k] = [s[k] for s in samples]
        return out


def get_vaild_loader(csv_path, dataset_path, support_img_size=384, text_tokenizer=None, batch_size=8, shuffle=False, num_workers=12,
                     pin_memory=True, prefetch_factor=4, worker_init_fn=None, device="cuda"):
    """ref: utils/dataloader.py:372-413 (same arguments; `device` is where the collate step runs the resize kernels).
    pin_memory is accepted for signature compatibility: the collate step already produces device tensors."""
    dataset = VaildingDataset(csv_path, dataset_path, support_img_size=support_img_size, text_tokenizer=text_tokenizer)
    kw = dict(dataset=dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, worker_init_fn=worker_init_fn,
              collate_fn=_HostCollate())
    if num_workers > 0:
        kw["prefetch_factor"] = prefetch_factor
    return _DeviceLoader(DataLoader(**kw), collate_on_device(device, support_img_size, dataset.query_img_size))


class _HostCollate:
    """Workers hand over the decoded samples as a plain list (variable image sizes cannot be stacked on the host)."""

    def __call__(self, samples):
        return samples


class _DeviceLoader:
    """Iterates the host DataLoader and runs the device collate in the consuming (main) process."""

    def __init__(self, loader, collate):
        self.loader, self.collate, self.dataset = loader, collate, loader.dataset

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for samples in self.loader:
            yield self.collate(samples)


def gallery_batches(csv_path, dataset_path, batch_size=8, device="cuda", num_workers=0):
    """Offline gallery builder input: yields {"query_img", "query_mask"} device batches for
    cor_amd.retrieval.build_gallery from the same CSV schema (one gallery row per CSV pair: query image + query mask)."""
    class _G(Dataset):
        def __init__(self):
            self.df = read_pairs_csv(csv_path)

        def __len__(self):
            return len(self.df)

        def __getitem__(self, i):
            row = self.df.iloc[i]
            root = os.path.join(dataset_path, row["Dataset"])
            return {"q": VaildingDataset.rgb_loader(os.path.join(root, "image", row["Query_img"])),
                    "m": VaildingDataset.binary_loader(os.path.join(root, "mask", str(row["Target"]), row["Query_mask"]))}

    dev = torch.device(device)
    q_img, q_mask = P.ImageTransform(1024), P.MaskTransform(1024)
    up = lambda a: torch.from_numpy(np.array(a, copy=True)).to(dev)                   # noqa: E731
    for samples in DataLoader(_G(), batch_size=batch_size, shuffle=False, num_workers=num_workers, collate_fn=_HostCollate()):
        yield {"query_img": torch.stack([q_img(up(s["q"])) for s in samples]),
               "query_mask": torch.stack([q_mask(up(s["m"])) for s in samples])}
