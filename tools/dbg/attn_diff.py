#!/usr/bin/env python3
"""Debug: where does variant 0 (w64) differ from variant 2 (pipe)? python tools/dbg/attn_diff.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cor_amd import ops
from cor_amd._native import Q_PRESCALE_HD64 as QC
torch.manual_seed(0)
B, H = 1, 2
d = H * 64
dev = "cuda:0"
qkv = torch.randn((B * 4096, 3 * d), device=dev).to(torch.bfloat16)
pad = torch.randn((3 * d,), device=dev).to(torch.bfloat16)
rh = torch.randn((127, 64), device=dev) * 0.3
rw = torch.randn((127, 64), device=dev) * 0.3
a = ops.sam_attention(qkv, pad, rh, rw, B, H, 64, 0, out_dtype=torch.float32, q_prescale=QC, variant=int(sys.argv[1]) if len(sys.argv) > 1 else 0).view(4096, H, 64)
b = ops.sam_attention(qkv, pad, rh, rw, B, H, 64, 0, out_dtype=torch.float32, q_prescale=QC, variant=int(sys.argv[2]) if len(sys.argv) > 2 else 2).view(4096, H, 64)
err = (a - b).abs()
bad = err > 0.05
print("bad elements", int(bad.sum()), "of", bad.numel(), "max", float(err.max()))
for h in range(H):
    rows = bad[:, h].any(1).nonzero().flatten()
    print("head", h, "bad rows", len(rows), "first", rows[:40].tolist())
    if len(rows):
        r0 = int(rows[0])
        print(" row", r0, "bad cols", bad[r0, h].nonzero().flatten().tolist())
        print(" new", a[r0, h, :8].tolist()); print(" old", b[r0, h, :8].tolist())
    # by (grid row, column) pattern
    br = bad[:, h].any(1).view(64, 64)
    print(" bad per grid-row:", br.sum(1).tolist())
    print(" bad per grid-col:", br.sum(0).tolist())
