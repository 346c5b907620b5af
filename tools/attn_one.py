#!/usr/bin/env python3
"""One configuration of the SAM attention kernels, a few launches (for rocprofv3 --pmc passes):
python tools/attn_one.py <variant: 0 default, 1 chain form, 2 fma-bias form, 4 64-query-per-wave form> [B] [window] [prescaled: 1 (default, what the engine runs) | 0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native
variant = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 32; window = int(sys.argv[3]) if len(sys.argv) > 3 else 0
H, g, dev, T = 12, 64, "cuda:0", torch.bfloat16
d = H * 64
S = 64 if window == 0 else 14
qkv = torch.randn((B * g * g, 3 * d), device=dev).to(T); pad = torch.randn((3 * d,), device=dev).to(T)
rh = torch.randn((2 * S - 1, 64), device=dev) * 0.5; rw = torch.randn((2 * S - 1, 64), device=dev) * 0.5
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 1
qp = _native.Q_PRESCALE_HD64 if pre else 1.0
if pre:
    qkv = qkv.float(); qkv[:, :d] *= qp; qkv = qkv.to(T)
for i in range(3): ops.sam_attention(qkv, pad, rh, rw, B, H, g, window, variant=variant, q_prescale=qp)
torch.cuda.synchronize()
