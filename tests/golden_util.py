"""Shared helpers for golden fixtures: seeded inputs (numpy PCG64, machine-stable), moments, sub-sampling."""
from __future__ import annotations

import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rect_mask(rng, B, size):
    """One random axis-aligned rectangle of ones per sample, area 5-50 %."""
    m = np.zeros((B, 1, size, size), dtype=np.float32)
    for b in range(B):
        frac = rng.uniform(0.05, 0.5)
        ar = rng.uniform(0.5, 2.0)
        h = int(np.clip(np.sqrt(frac * ar) * size, 2, size))
        w = int(np.clip(np.sqrt(frac / ar) * size, 2, size))
        y0 = int(rng.integers(0, size - h + 1))
        x0 = int(rng.integers(0, size - w + 1))
        m[b, 0, y0:y0 + h, x0:x0 + w] = 1.0
    return m


def text_tokens(rng, B, ctx=64, vocab=32000):
    """4-20 random ids in [2, vocab) then pad id 1 (SigLIP pads with 1; the last position is pooled)."""
    t = np.ones((B, ctx), dtype=np.int64)
    for b in range(B):
        n = int(rng.integers(4, 21))
        t[b, :n] = rng.integers(2, vocab, size=n)
    return t


def make_inputs(seed, **spec):
    """spec values: shape tuple -> N(0,1) fp32 ; ("mask", B, size) ; ("tokens", B, ctx, vocab)."""
    rng = np.random.default_rng(seed)
    out = {}
    for k in sorted(spec):
        v = spec[k]
        if isinstance(v[0], str):
            if v[0] == "mask":
                a = rect_mask(rng, v[1], v[2])
            elif v[0] == "tokens":
                a = text_tokens(rng, v[1], v[2], v[3])
            else:
                raise ValueError(v[0])
        else:
            a = rng.standard_normal(v, dtype=np.float32)
        out[k] = torch.from_numpy(a)
    return out


def moments(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.pow(2).mean().sqrt().item()], dtype=np.float64)


def strided(t, s):
    return t.detach()[..., ::s, ::s].contiguous().numpy()


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
