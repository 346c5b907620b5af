"""SigLIP text tokenizer from a LOCAL sentencepiece model file.

The reference calls open_clip.get_tokenizer(model_name) (lib/support_model/siglip_openclip.py:15, utils/dataloader.py:296), which
downloads an HF-hub tokenizer by name; no vocabulary file ships with the reference and there is no network here. This restates
open_clip's SigLipTokenizer (open_clip_torch 2.31.0, third party, UNVERIFIED offline): canonicalise (lower-case, strip punctuation,
collapse whitespace), sentencepiece-encode, append EOS (id 1), truncate / pad to `context_length` with the pad id (1 for the
SigLIP c4-en vocabulary: the last position, which the text tower pools, is then the pad/EOS token)."""
from __future__ import annotations

import re
import string

import torch


def canonicalize_text(text: str) -> str:
    text = text.translate(str.maketrans("", "", string.punctuation)).lower()
    return re.sub(r"\s+", " ", text).strip()


def siglip_tokenizer(spm_model_path: str, context_length: int = 64, pad_id: int = 1, eos_id: int = 1):
    import sentencepiece as spm
    sp = spm.SentencePieceProcessor(model_file=spm_model_path)

    def tokenize(text: str) -> torch.Tensor:
        ids = sp.encode(canonicalize_text(text))[: context_length - 1] + [eos_id]
        out = torch.full((context_length,), pad_id, dtype=torch.int64)
        out[: len(ids)] = torch.tensor(ids, dtype=torch.int64)
        return out

    return tokenize


def hashing_tokenizer(vocab: int = 32000, context_length: int = 64, pad_id: int = 1):
    """Deterministic stand-in for smoke runs without a vocabulary file: word -> 2 + hash(word) % (vocab - 2)."""
    import zlib

    def tokenize(text: str) -> torch.Tensor:
        ids = [2 + zlib.crc32(w.encode()) % (vocab - 2) for w in canonicalize_text(text).split()][: context_length - 1]
        out = torch.full((context_length,), pad_id, dtype=torch.int64)
        out[: len(ids)] = torch.tensor(ids, dtype=torch.int64)
        return out

    return tokenize
