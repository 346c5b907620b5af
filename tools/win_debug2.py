#!/usr/bin/env python3
"""Which LDS slots of V reach the output? q = 0 (uniform attention), V[token][d] = 1 iff (slot >> 2) == d, slot = 16*kh + kw."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops
DEV = "cuda:0"; BF16 = torch.bfloat16
B, H, grid = 1, 1, 28
d = H * 64
qkv = torch.zeros((B * grid * grid, 3, d), device=DEV)
qkv[:, 1] = torch.randn((B * grid * grid, d), device=DEV)      # K random (irrelevant: q = 0)
yy, xx = torch.meshgrid(torch.arange(grid), torch.arange(grid), indexing="ij")
slot = ((yy % 14) * 16 + (xx % 14)).reshape(-1).to(DEV)
qkv[torch.arange(grid * grid, device=DEV), 2, slot >> 2] = 1.0
qkv = qkv.reshape(-1, 3 * d).to(BF16)
pad = torch.zeros((3 * d,), device=DEV).to(BF16)
rh = torch.zeros((27, 64), device=DEV); rw = torch.zeros((27, 64), device=DEV)
for variant in (1, 0):
    out = ops.sam_attention(qkv, pad, rh, rw, B, H, grid, 14, out_dtype=torch.float32, variant=variant).cpu() * 196
    o = out.view(grid, grid, 64)
    print("variant", variant, "distinct output rows:", len(torch.unique(o.reshape(-1, 64).round(decimals=2), dim=0)))
    for (y, x) in ((0, 0), (5, 9), (13, 13), (20, 3)):
        print("  token", (y, x), [round(float(v), 1) for v in o[y, x, :56]])
    ref = o[0, 0]
    badmask = ((o - ref).abs().max(dim=-1).values > 0.01)
    ys, xs = badmask.nonzero(as_tuple=True)
    print("  bad tokens:", [(int(y), int(x), (int(y) % 14) * 14 + int(x) % 14) for y, x in zip(ys[:40], xs[:40])], "count", int(badmask.sum()))
    if len(ys):
        y, x = int(ys[0]), int(xs[0])
        print("  first bad row", [round(float(v), 1) for v in o[y, x, :56]])
