#!/usr/bin/env python3
"""Timing ablations of flash_global_w64 (probe library: COR_AMD_LIB=tools/probes/libcor_probes.so). python tools/dbg/attn_ablate.py [B]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cor_amd import ops
from cor_amd._native import Q_PRESCALE_HD64 as QC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, g, dev, T = 12, 64, "cuda:0", torch.bfloat16
d = H * 64
qkv = torch.randn((B * g * g, 3 * d), device=dev)
qkv[:, :d] *= QC
qkv = qkv.to(T)
pad = torch.randn((3 * d,), device=dev).to(T)
rh = torch.randn((127, 64), device=dev) * 0.5
rw = torch.randn((127, 64), device=dev) * 0.5
names = {0: "w64", 2: "pipe", 17: "-copies", 18: "-wait/barrier", 19: "-copies-barrier", 20: "-fragment reads", 23: "-copies-barrier-reads", 24: "-exp2", 32: "-max",
         48: "-PV mfma", 80: "-QK mfma", 112: "-all mfma", 143: "-everything", 144: "-rowsum", 272: "-sub", 400: "-rowsum-sub", 3: "pipe+CB", 528: "w64 copies at top", 4: "w64", 1040: "w64 register staging"}
variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(names)
for rnd in range(2):
    for v in variants:
        f = lambda: ops.sam_attention(qkv, pad, rh, rw, B, H, g, 0, variant=v, q_prescale=QC)
        for i in range(25): f()
        ts = []
        for i in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(json.dumps(dict(round=rnd, variant=v, name=names.get(v, "?"), ms=round(min(ts), 4))), flush=True)
