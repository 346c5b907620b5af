#!/usr/bin/env python3
"""sha256 of the global / windowed SAM attention outputs on fixed seeded inputs (two builds of the library are compared bit for bit by
running this under COR_AMD_LIB=<other build>). python tools/attn_hash.py [B]"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops
from cor_amd._native import Q_PRESCALE_HD64 as QC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, g, dev, T = 12, 64, "cuda:0", torch.bfloat16
gen = torch.Generator(device="cpu").manual_seed(7)
d = H * 64
out = {}
for amp in (1.0, 4.0):                                   # amp 4: sharp rows, the reference moves more often
    qkv = (torch.randn((B * g * g, 3 * d), generator=gen) * amp).to(T).to(dev)
    pad = torch.randn((3 * d,), generator=gen).to(T).to(dev)
    for window, S in ((0, 64), (14, 14)):
        rh = (torch.randn((2 * S - 1, 64), generator=gen) * 0.5).to(dev)
        rw = (torch.randn((2 * S - 1, 64), generator=gen) * 0.5).to(dev)
        o = ops.sam_attention(qkv, pad, rh, rw, B, H, g, window, q_prescale=QC)
        torch.cuda.synchronize()
        out[f"amp{amp}_win{window}"] = hashlib.sha256(o.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16]
        assert torch.isfinite(o.float()).all()
print(json.dumps(out))
