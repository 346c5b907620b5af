#!/usr/bin/env python3
"""GPU micro-benchmark of the SAM attention kernels (global / windowed) at SAM-B shape. python tools/attn_bench.py [B]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, g, dev, T = 12, 64, "cuda:0", torch.bfloat16
d = H * 64
qkv = torch.randn((B * g * g, 3 * d), device=dev).to(T)
pad = torch.randn((3 * d,), device=dev).to(T)
from cor_amd._native import Q_PRESCALE_HD64 as QC
for window, S, variant, qp in ((0, 64, 0, QC), (0, 64, 2, QC), (0, 64, 4, QC), (0, 64, 1, 1.0), (0, 64, 0, QC), (0, 64, 2, QC), (0, 64, 4, QC), (0, 64, 1, QC), (14, 14, 0, QC), (14, 14, 1, 1.0), (14, 14, 0, QC)):   # variant / q_prescale: per-call choices
    rh = torch.randn((2 * S - 1, 64), device=dev) * 0.5
    rw = torch.randn((2 * S - 1, 64), device=dev) * 0.5
    qk = qkv
    if qp != 1.0:                                        # the caller folds the factor into q (as engine.pack does): same logits as the raw run
        qk = qkv.float(); qk[:, :d] *= qp; qk = qk.to(T)
        pd = pad.float(); pd[:d] *= qp; pd = pd.to(T)
    else:
        pd = pad
    for i in range(40 if window == 0 else 200):          # ~100 ms of the same kernel first: clocks settle, caches warm
        ops.sam_attention(qk, pd, rh, rw, B, H, g, window, variant=variant, q_prescale=qp)
    ts = []
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.sam_attention(qk, pd, rh, rw, B, H, g, window, variant=variant, q_prescale=qp); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    fl = (4.0 * (g * g) ** 2 * 64 if window == 0 else 25 * 4.0 * 196 ** 2 * 64) * H * B
    print(json.dumps(dict(window=window, variant=variant, q_prescale=round(qp, 4), B=B, ms=min(ts[1:]), tflops=fl / (min(ts[1:]) * 1e-3) / 1e12)), flush=True)
